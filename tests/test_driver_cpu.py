"""Host logic of the train.py counterpart: window slicing, length clipping,
loss accumulation.  Pure index arithmetic: runs on CPU."""
import numpy as np

from multinn_amd.driver import iter_windows, LossAccumulator, TrainingStats
from oracle import generators as G


def test_iter_windows_matches_reference_loop():
    lengths = np.array([10, 3, 7, 12, 0, 5])
    ids = np.array([3, 0, 5, 1, 4, 2])
    got = [w for w in iter_windows(ids, lengths, 12, batch_size=4, piece_size=4)]
    # hand evaluation of train.py:164-173
    exp = []
    for i in range(0, 6, 4):
        b = ids[i:i + 4]
        for j in range(0, 12, 4):
            lb = lengths[b] - j
            ne = np.where(lb > 0)[0]
            if len(ne):
                exp.append((list(b[ne]), j, int(np.minimum(lb[ne], 4).max()), list(np.minimum(lb[ne], 4))))
        exp.append(None)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        if e is None:
            assert g is None
        else:
            assert (list(g[0]), g[1], g[2], list(g[3])) == e
    # same row counts as the oracle's window list (K18)
    w = G.training_windows(10, [10, 3, 7], 4)
    mine = [x for x in iter_windows(np.arange(3), np.array([10, 3, 7]), 10, 3, 4) if x is not None]
    assert [int(a[3].sum()) for a in mine] == [int(b[3].sum()) for b in w]
    assert sum(1 for x in got if x is None) == 2            # stats.new_step() once per song batch, not per piece


def test_loss_accumulator_and_stats(tmp_path):
    acc = LossAccumulator()
    for v in (1.0, float("nan"), 3.0, float("inf"), -float("inf")):
        acc.update(v)
    assert acc.loss() == 2.0 and acc.num_bad() == 3 and abs(acc.ratio_bad() - 0.6) < 1e-12
    assert "nan: 1" in str(acc) and "bad: 60.00%" in str(acc)
    acc.clear()
    assert np.isnan(acc.loss())
    s = TrainingStats()
    s.new_step(); s.new_epoch(); s.update_metric_best(1.5)
    s.save(str(tmp_path / "steps"))
    t = TrainingStats()
    assert t.load(str(tmp_path / "steps")) and (t.steps, t.epoch, t.metric_best, t.run) == (1, 1, 1.5, 2)
    assert not TrainingStats().load(str(tmp_path / "missing"))
