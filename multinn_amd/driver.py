"""train.py:138-282 counterpart: the batching / windowing loop, loss accumulation,
best/last checkpoint policy and early stopping around ``Generator.train_step``.

Same YAML keys as multinn/configs/default_config.yaml / default_params.yaml and the same CLI flags as
train.py:287-329 (``python -m multinn_amd.driver -m NAME -c CONFIG -p PARAMS``).  Every mode of
``params['mode']`` is wired through ``multinn_amd.modes`` (``build_model``); joint + PassEncoder + NADE takes the
fused, hipGraph-replayed RnnNade step directly.
"""
import argparse
import math
import os
import pickle
import sys
import time

import numpy as np


class LossAccumulator:
    """Running mean of the per-batch losses of an epoch that keeps non-finite values OUT of the mean and tallies them by kind (the
    behaviour train.py:150-195 relies on from utils/training.py:98-148: a NaN batch is reported, not averaged)."""

    KINDS = ("nan", "+inf", "-inf")

    def __init__(self):
        self.clear()

    def clear(self):
        self.finite_sum, self.finite_n = 0.0, 0
        self.bad = dict.fromkeys(self.KINDS, 0)

    @staticmethod
    def _kind(x):
        if math.isnan(x):
            return "nan"
        if math.isinf(x):
            return "+inf" if x > 0 else "-inf"
        return None

    def update(self, loss):
        loss = float(loss)
        kind = self._kind(loss)
        if kind is None:
            self.finite_sum += loss
            self.finite_n += 1
        else:
            self.bad[kind] += 1

    def loss(self):
        return self.finite_sum / self.finite_n if self.finite_n else float("nan")

    def num_bad(self):
        return sum(self.bad.values())

    def ratio_bad(self):
        seen = self.finite_n + self.num_bad()
        return self.num_bad() / seen if seen else 0.0

    def __str__(self):
        tally = ", ".join(f"{k}: {n}" for k, n in self.bad.items())
        return f" - loss: {self.loss():7.3f} ({tally}, bad: {100.0 * self.ratio_bad():.2f}%)"


class TrainingStats:
    """utils/training.py:12-95: (steps, epoch, run, metric_best) pickled next to the checkpoint."""

    def __init__(self):
        self.steps, self.epoch, self.run, self.metric_best, self.idle_epochs = 0, 0, 1, float("inf"), 0

    def new_step(self):
        self.steps += 1

    def new_epoch(self):
        self.epoch += 1

    def new_idle_epoch(self):
        self.idle_epochs += 1

    def reset_idle_epochs(self):
        self.idle_epochs = 0

    def update_metric_best(self, m):
        self.metric_best = m

    def save(self, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "wb") as f:
            pickle.dump((self.steps, self.epoch, self.run, self.metric_best), f)

    def load(self, path):
        if not os.path.exists(path):
            return False
        with open(path, "rb") as f:
            self.steps, self.epoch, self.run, self.metric_best = pickle.load(f)
        self.run += 1
        return True


def iter_windows(ids, lengths, T_total, batch_size, piece_size):
    """train.py:164-173: for every batch of shuffled song ids and every piece of ``piece_size`` steps yield
    (song_ids, j, max_length, clipped_lengths) for the songs that still have steps (empty songs dropped)."""
    lengths = np.asarray(lengths)
    for i in range(0, len(ids), batch_size):
        batch = ids[i:i + batch_size]
        for j in range(0, T_total, piece_size):
            len_batch = lengths[batch] - j
            non_empty = np.where(len_batch > 0)[0]
            if len(non_empty) > 0:
                lb = np.minimum(len_batch[non_empty], piece_size)
                yield batch[non_empty], j, int(lb.max()), lb.astype(np.int32)
        yield None                                    # end of a song batch: stats.new_step() (train.py:194)


def train_epoch(generator, X, lengths, ids, batch_size, piece_size, optimizer, loss_accum, stats, lr=None, device=None):
    """One epoch of train.py:164-194.  X uint8 [S,T,P,M] (numpy); every window starts from a ZERO rnn state."""
    import torch
    for w in iter_windows(ids, lengths, X.shape[1], batch_size, piece_size):
        if w is None:
            stats.new_step()
            continue
        song_ids, j, max_len, len_batch = w
        xb = torch.from_numpy(np.ascontiguousarray(X[song_ids, j:j + max_len])).to(device or "cuda")
        full = bool((len_batch == max_len).all())
        lb = None if full else torch.from_numpy(len_batch).to(xb.device)
        run = _captured_step(generator, xb, optimizer, lr, lengths=lb)
        if run is not None:
            loss = run(xb) if full else run(xb, lb)
        else:
            loss = generator.train_step(xb, lb, optimizer, lr)
        loss_accum.update(float(loss))
    return loss_accum.loss()


def _captured_step(generator, xb, optimizer, lr, max_graphs=8, lengths=None):
    """An eager optimiser step is host-bound (C2 shape: 7.2 ms eager, 2.85 ms as a hipGraph replay), so windows of a shape that keeps
    coming back -- batch_size x piece_size, i.e. nearly all of an epoch -- run as replays of RnnNade.graphed_train_step.
    A shape is captured at its SECOND occurrence (the first one runs eagerly and creates every workspace; capturing executes nothing,
    so the trajectory is the eager one), keyed by optimiser and learning rate (both are baked into the graph).  Only for the paths the
    captured step is tested on: the two-layer persistent recurrence and the row-parallel (CU-resident / cluster) one.
    RAGGED windows (lengths given) are captured too where the generator runs them compacted (16-bit RnnNade: every row count lives on the
    device, so ONE graph per window shape serves any lengths -- the reference's data is ragged, train.py:165-173); they get their own graph
    beside the full-length one of the same shape, which skips the compaction passes.  MULTINN_TRAIN_GRAPH=0 keeps every step eager."""
    import os
    if os.environ.get("MULTINN_TRAIN_GRAPH", "1") == "0" or not xb.is_cuda or not hasattr(generator, "graphed_train_step"):
        return None
    ragged = lengths is not None
    if getattr(generator, "_mode", None) in ("joint", "jamming", "composer") and hasattr(generator, "generators"):
        # a mode class (multinn_amd.modes): its own captured step -- encoders, every generator, the joint clipped step -- full-length or ragged
        # (MultINNCore.graphed_train_step keeps every row count of a ragged window on the device)
        from .training import dp_active
        if dp_active() or any(getattr(g, "store", None) is None or g.store.theta is None for g in generator.generators):
            return None
    else:
        stack = getattr(generator, "_stack", None)
        if stack is None or getattr(stack, "packed", None) is None or not (stack._persist(xb.shape[0], xb.shape[1]) or stack._rowpar(xb.shape[0], xb.shape[1])):
            return None
        if ragged:
            from . import ops as _ops
            if not (getattr(generator, "ragged_compact", False) and getattr(generator, "dtype", None) in _ops.H16 and getattr(generator, "num_tracks", 0) == 1):
                return None
    key = (tuple(xb.shape), id(optimizer), lr, "ragged" if ragged else "full")
    graphs = generator.__dict__.setdefault("_step_graphs", {})
    if key in graphs:
        return graphs[key]
    seen = generator.__dict__.setdefault("_step_shapes_seen", set())
    if key not in seen or len(graphs) >= max_graphs:
        seen.add(key)
        return None
    graphs[key] = generator.graphed_train_step(xb, optimizer, lr, warmup=0, lengths=lengths)
    return graphs[key]


def evaluate(generator, X, lengths, batch_size, piece_size, device=None):
    """utils/training.py:180-213 collect_metrics: mean generator NLL per valid row over all windows."""
    import torch
    tot, cnt = 0.0, 0
    ids = np.arange(X.shape[0])
    for w in iter_windows(ids, lengths, X.shape[1], batch_size, piece_size):
        if w is None:
            continue
        song_ids, j, max_len, len_batch = w
        xb = torch.from_numpy(np.ascontiguousarray(X[song_ids, j:j + max_len])).to(device or "cuda")
        full = bool((len_batch == max_len).all())
        generator.build_pianoroll(xb, None if full else torch.from_numpy(len_batch).to(xb.device), is_train=False, mode="eval")
        n = int(len_batch.sum())
        loss = generator.generator_loss() if hasattr(generator, "generator_loss") else generator.metrics["batch/loss"]
        tot += float(loss) * n
        cnt += n
    return tot / max(cnt, 1)


def check_recurrences(model):
    """Raise if any persistent-recurrence launch of the model's LSTM stacks timed out on a bounded spin since the last check (its sticky
    status word: LstmStack.check / MultINNCore.check; synchronises the device).  The launches of lstm_persist.hip / lstm_rowpar.hip need
    their whole grid resident at once; one that could not become resident returns garbage and only sets that word."""
    if hasattr(model, "check"):
        try:
            model.check(tolerate_overflow=True)          # the training loop: f16 overflows are answered by the dynamic loss scale (a warning)
        except TypeError:
            model.check()
    elif getattr(model, "_stack", None) is not None:
        model._stack.check()


def fit(generator, optimizer, X_train, len_train, X_valid, len_valid, training_config, logs_config, dirs, stats=None, beat_size=4,
        save_best_only=False, log=print):
    """train.py:150-282: epochs, shuffling (np.random.seed(epoch)), evaluation, best / last checkpoints,
    early stopping after ``early_stopping`` idle epochs."""
    stats = stats or TrainingStats()
    batch_size = training_config["batch_size"]
    piece_size = int(training_config["piece_size"] * beat_size)
    ids = np.arange(X_train.shape[0])
    acc = LossAccumulator()
    past = stats.epoch
    for epoch in range(past + 1, past + training_config["epochs"] + 1):
        stats.new_epoch()
        np.random.seed(epoch)
        t0 = time.time()
        np.random.shuffle(ids)
        acc.clear()
        train_epoch(generator, X_train, len_train, ids, batch_size, piece_size, optimizer, acc, stats, training_config.get("learning_rate"))
        check_recurrences(generator)            # a persistent launch that gave up would have trained on garbage: raise before validating / saving
        loglik_val = evaluate(generator, X_valid, len_valid, batch_size, piece_size) if logs_config.get("evaluate_epochs", 1) else acc.loss()
        log(f" epoch: {epoch:3d} (steps: {stats.steps:5d}) time: {time.time() - t0:.2f}s{acc} valid nll: {loglik_val:.4f}")
        if loglik_val < stats.metric_best:
            stats.update_metric_best(loglik_val)
            stats.reset_idle_epochs()
            if logs_config.get("save_checkpoint_epochs", 1) > 0 and epoch % logs_config.get("save_checkpoint_epochs", 1) == 0:
                generator.save(None, dirs["model_dir"], global_step=stats.steps)
                stats.save(os.path.join(dirs["model_dir"], "steps"))
        else:
            stats.new_idle_epoch()
            if stats.idle_epochs >= training_config["early_stopping"]:
                log(f"[WARN]  No improvement after {training_config['early_stopping']} epochs, quiting")
                break
    check_recurrences(generator)
    if not save_best_only:
        generator.save(None, dirs["model_last_dir"], global_step=stats.steps)
        stats.save(os.path.join(dirs["model_last_dir"], "steps"))
    return stats


def pretrain_encoders(encoders, X_train, len_train, training_config, beat_size=4, lr=None, device=None, log=None):
    """train_encoders.py:118-200: greedy layer-wise pre-training of the per-track (or joint) DBN encoders.  For every layer, for
    `epochs` epochs, every shuffled song batch and every piece of `piece_size` beats: the encoder is built on the window of its
    track (`x[..., i]` for track i, all tracks flattened for a single joint encoder) and `encoder.train(None, lr, layer)` runs one
    CD-k update of that layer's RBM fed with the sampled codes of the layers below (dbn_encoder.py:192-240).
    X_train uint8 [songs, T, P, M] (numpy).  Returns {(encoder index, layer): [mean reconstruction cost per epoch]}."""
    import torch
    batch_size = training_config['batch_size']
    piece_size = int(training_config['piece_size'] * beat_size)
    lr = training_config['learning_rate'] if lr is None else lr
    ids = np.arange(X_train.shape[0])
    history = {}
    for e_i, enc in enumerate(encoders):
        for layer in range(enc.num_layers):
            costs = []
            for epoch in range(1, training_config['epochs'] + 1):
                np.random.seed(epoch)                           # train_encoders.py:134-135
                np.random.shuffle(ids)
                acc = LossAccumulator()
                for w in iter_windows(ids, len_train, X_train.shape[1], batch_size, piece_size):
                    if w is None:
                        continue
                    song_ids, j, max_len, _ = w
                    xb = X_train[song_ids, j:j + max_len]
                    xb = xb[..., e_i] if len(encoders) > 1 else xb.reshape(xb.shape[0], xb.shape[1], -1)
                    xt = torch.from_numpy(np.ascontiguousarray(xb)).to(device or "cuda")
                    enc.build(x=xt, is_train=True, mode="train")
                    _, _, metrics, _, _ = enc.train(None, lr, layer=layer)
                    acc.update(float(metrics["batch/loss"]) if metrics else 0.0)
                costs.append(acc.loss())
                if log is not None:
                    log(f"[PRETRAIN] encoder {e_i} layer {layer} epoch {epoch}: reconstruction cost {costs[-1]:.4f}")
            history[(e_i, layer)] = costs
    return history


def build_generator(params, P, M, precision="bf16", seed=23):
    """multinn_joint.py:41-74 for `mode: joint`, `encoder.type: Pass`, `generator.type: NADE|RBM`: the bare generator (kept for
    callers that drive a Generator directly; `build_model` wires every mode)."""
    from .generators import RnnNade, RnnRBM
    if params.get("mode", "joint") != "joint" or (params.get("encoder") or {}).get("type", "Pass") != "Pass":
        raise ValueError("build_generator is the joint / PassEncoder shortcut: use build_model(config, params) for the other modes")
    g = params["generator"]
    cls = {"NADE": RnnNade, "RBM": RnnRBM}[g["type"]]
    return cls(P * M, g["num_hidden"], g["num_hidden_rnn"], keep_prob=params.get("keep_prob", 0.9), precision=precision, seed=seed)


def build_model(config, params, precision="bf16", seed=None, device=None):
    """train.py:50: `MultINN(config, params, mode=params['mode'], name=config['model_name'])` -- any of the five modes
    (multinn_amd.modes), with the train_step / build_pianoroll / save / load surface `fit` drives."""
    from .modes import MultINN
    return MultINN(config, params, mode=params["mode"], name=config.get("model_name", "MultINN"), precision=precision, seed=seed, device=device)


def main(argv=None):
    import yaml
    ap = argparse.ArgumentParser(description="MultINN joint LSTM-NADE training on MI355X (flags of multinn/train.py)")
    ap.add_argument("--model-name", "-m", required=True)
    ap.add_argument("--config", "-c", default="configs/default_config.yaml")
    ap.add_argument("--params", "-p", default="configs/default_params.yaml")
    ap.add_argument("--epochs", "-e", type=int, default=None)
    ap.add_argument("--learning-rate", "--lr", type=float, default=None)
    ap.add_argument("--from-init", action="store_true")
    ap.add_argument("--from-last", action="store_true")
    ap.add_argument("--encoders", default=None)
    ap.add_argument("--save-best-only", action="store_true")
    ap.add_argument("--reuse-config", action="store_true")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"],
                    help="fp16: IEEE-half operands, the mode that meets the 1e-4 parity gate at full speed; bf16; fp32: v_mfma_f32 GEMMs, launch-per-step recurrence")
    a = ap.parse_args(argv)
    from .training import AdamOptimizer
    root = os.path.join("..", "results", a.model_name)                  # utils/setup.py:50
    dirs = dict(model_dir=os.path.join(root, "ckpt", "model"), model_last_dir=os.path.join(root, "ckpt", "model", "last"),
                config_dir=os.path.join(root, "config"))
    if a.reuse_config:
        a.config, a.params = os.path.join(dirs["config_dir"], "config.yaml"), os.path.join(dirs["config_dir"], "params.yaml")
    config, params = yaml.safe_load(open(a.config)), yaml.safe_load(open(a.params))
    if a.epochs is not None:
        config["training"]["epochs"] = a.epochs
    if a.learning_rate is not None:
        config["training"]["learning_rate"] = a.learning_rate
    os.makedirs(dirs["config_dir"], exist_ok=True)
    yaml.safe_dump(config, open(os.path.join(dirs["config_dir"], "config.yaml"), "w"))
    yaml.safe_dump(params, open(os.path.join(dirs["config_dir"], "params.yaml"), "w"))
    d, tr = config["data"], config["training"]
    X = np.load(d["filename"] if d["filename"].endswith(".npy") else d["filename"] + ".npy").astype(np.uint8)
    npx = tr.get("num_pixels", 1)
    if npx > 1:                                                            # utils/data.py:71-79
        S, T, P, M = X.shape
        X = X[:, :T // npx * npx].reshape(S, T // npx, npx, P, M).transpose(0, 1, 3, 2, 4).reshape(S, T // npx, P * npx, M)
    lengths = np.full(X.shape[0], X.shape[1], np.int64) if not d.get("sequence_lengths") else np.load(d["sequence_lengths"]) // npx
    nt, nv = d["split"]["num_train"], d["split"]["num_valid"]
    if params.get("mode", "joint") == "joint" and (params.get("encoder") or {}).get("type", "Pass") == "Pass" \
            and params["generator"]["type"] == "NADE":
        gen = build_generator(params, X.shape[2], X.shape[3], a.precision, tr["random_seed"])     # the fused, hipGraph-replayed step
        gen._materialize(X.shape[2] * X.shape[3])
    else:
        config.setdefault("data", {}).setdefault("pitch_range", {"lowest": 0, "highest": X.shape[2] // max(npx, 1)})
        gen = build_model(config, params, a.precision, tr["random_seed"])
    stats = TrainingStats()
    if not a.from_init:
        src = dirs["model_last_dir"] if a.from_last else dirs["model_dir"]
        if gen.load(None, src):
            stats.load(os.path.join(src, "steps"))
    fit(gen, AdamOptimizer(tr["learning_rate"]), X[:nt], lengths[:nt], X[nt:nt + nv], lengths[nt:nt + nv], tr, config["logs"], dirs, stats,
        beat_size=d["beat_resolution"] / npx, save_best_only=a.save_best_only)


if __name__ == "__main__":
    main()
