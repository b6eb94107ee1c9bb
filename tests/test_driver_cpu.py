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


def test_split_k_heuristic_targets_workgroup_counts():
    """LstmStack._split_k: about 512 workgroups of 128 x 128 tiles below K = 64 k (every slice adds its tile with f32 atomics, which run
    at one chip-wide rate), about 1024 above, never more slices than K / 1024, at least one."""
    from multinn_amd.generators import LstmStack
    sk = LstmStack._split_k
    assert sk(2048, 448, 32768) == 8 and sk(2048, 512, 32768) == 8          # C2 layer 1: 64 tiles
    assert sk(1024, 512, 32768) == 16 and sk(1024, 256, 32768) == 32        # C2 layer 2: 32 / 16 tiles
    assert sk(2048, 448, 262144) == 16 and sk(1024, 256, 262144) == 64      # TGT: twice the slices
    assert sk(2048, 448, 2048) == 2 and sk(64, 64, 512) == 1 and sk(128, 128, 100) == 1


def test_data_parallel_switch_needs_a_process_group(monkeypatch):
    """training.dp_active: false without an initialised process group, whatever the rehearsal variable says."""
    from multinn_amd import training
    monkeypatch.setenv("MULTINN_DP_REHEARSAL", "1")
    assert training.dp_active() is False
    import torch
    g = torch.ones(4)
    assert training.allreduce_flat(g) is g and float(g.sum()) == 4.0


def test_bench_cli_contract():
    """bench.py's flags as the driver passes them (no GPU touched: only the argument parser and the workload table)."""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    flags = {a.value for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "attr", "") == "add_argument"
             for a in n.args if isinstance(a, ast.Constant)}
    assert {"--gpus", "--steps", "--warmup"} <= flags
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key


def test_pack_epoch_dates_the_repacked_sampling_weights():
    """common.Model._packed_step: every "stale" mark (-1) starts a new pack epoch -- what LstmStack.det_job keys its repacked f32 sampling
    weights on (one repack per sampling scan, inside its captured graph); recording the store step the 16-bit packs were made at does not."""
    from multinn_amd.common import Model

    class M(Model):
        def build(self, *a, **k):
            pass

        def build_metrics(self, *a, **k):
            return [], [], None

    m = M()
    assert m._packed_step == -1 and getattr(m, "_pack_epoch", 0) == 0
    m._packed_step = 7
    assert m._packed_step == 7 and getattr(m, "_pack_epoch", 0) == 0
    m._packed_step = -1
    m._packed_step = -1
    assert m._packed_step == -1 and m._pack_epoch == 2


def test_store_check_tolerates_fp16_overflow_until_the_scale_is_at_its_floor():
    """ParamStore.check: strict by default (any skipped optimiser step raises); a training loop in precision "fp16" passes tolerate_overflow --
    skipped steps are then what the dynamic loss scale feeds on (a warning), until the multiplier has been halved down to its floor (a NaN)."""
    import warnings
    import pytest
    import torch
    from multinn_amd.common import ParamStore
    st = ParamStore(torch.device("cpu"))
    st.declare("w", (3, 4), None)
    st.materialize()
    st.check()                                               # nothing skipped
    st.skipped.fill_(2)
    with pytest.raises(FloatingPointError):
        st.check()
    assert int(st.skipped) == 0
    st.skipped.fill_(1)
    st.ls_dyn.copy_(torch.tensor([0.25, 4.0]))
    with pytest.warns(UserWarning, match="0.25"):
        st.check(tolerate_overflow=True)
    st.skipped.fill_(1)
    st.ls_dyn.copy_(torch.tensor([2.0 ** -20, 2.0 ** 20]))
    with pytest.raises(FloatingPointError):
        st.check(tolerate_overflow=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        st.check(tolerate_overflow=True)                     # counter cleared: silent


def test_store_checkpoint_carries_the_dynamic_loss_scale():
    import torch
    from multinn_amd.common import ParamStore

    def store():
        st = ParamStore(torch.device("cpu"))
        st.declare("w", (2, 3), None)
        st.materialize()
        return st

    a = store()
    a.ls_dyn.copy_(torch.tensor([0.125, 8.0])); a.ls_good.fill_(17); a.step_dev.fill_(5)
    sd = a.state_dict()
    b = store()
    b.load_state_dict(sd)
    assert b.ls_dyn.tolist() == [0.125, 8.0] and int(b.ls_good) == 17 and int(b.step_dev) == 5
    sd.pop("ls_dyn"); sd.pop("ls_good")                      # a checkpoint written before the dynamic scale existed
    c = store()
    c.load_state_dict(sd)
    assert c.ls_dyn.tolist() == [1.0, 1.0]
