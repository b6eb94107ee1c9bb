"""metrics/statistical.py:6-47 -- per-batch scalar metrics (host-side reporting, off the hot path).

``batch/loss`` (statistical.py:34) is the optimised loss; the streaming tf.metrics become running
(total, count) pairs updated by calling the returned update closures.  R5 (f1 uses precision for
recall) is reproduced; R6 (perplexity = exp of a D-dim sum) is computed in float64 to avoid the
reference's float32 overflow.
"""
import torch


class Running:
    def __init__(self):
        self.total, self.count = 0.0, 0.0

    def update(self, total, count):
        self.total += float(total)
        self.count += float(count)

    def value(self):
        return self.total / self.count if self.count else 0.0


def base_metrics(loss, targets, predictions, log_probs):
    t = targets.reshape(predictions.shape).bool()
    p = predictions.bool()
    tp, fp, fn = (t & p).sum(), (~t & p).sum(), (t & ~p).sum()
    state = {k: Running() for k in ("loss", "log_likelihood", "perplexity", "accuracy", "tp", "fp", "fn")}

    def upd():
        n = loss.numel()
        state["loss"].update(loss.sum(), n)
        state["log_likelihood"].update(log_probs.sum(), n)
        state["perplexity"].update(torch.exp(log_probs.double()).clamp(max=1e300).sum(), n)
        state["accuracy"].update((t == p).sum(), t.numel())
        state["tp"].update(tp, 1); state["fp"].update(fp, 1); state["fn"].update(fn, 1)

    upd()

    def prec():
        d = state["tp"].total + state["fp"].total
        return state["tp"].total / d if d else 0.0

    def rec():
        d = state["tp"].total + state["fn"].total
        return state["tp"].total / d if d else 0.0

    metrics = {
        "loss": state["loss"].value(), "log_likelihood": state["log_likelihood"].value(),
        "perplexity": state["perplexity"].value(), "accuracy": state["accuracy"].value(),
        "precision": prec(), "recall": rec(),
        "batch/loss": loss.mean(),
    }
    pr = metrics["precision"]
    metrics["f1_score"] = 2 * (pr * pr) / (pr + pr) if pr > 0 else 0.0        # statistical.py:37-38 (R5)
    return metrics, [upd], None
