"""Boundary checks that need no GPU: every plugin class is constructed with EXACTLY the keyword arguments of the
reference's own call sites, the mode classes wire the right encoder / generator classes for a `config` / `params`
pair, defaults match the reference, and the product path refuses CPU tensors (no CPU fallback)."""
import inspect

import numpy as np
import pytest
import torch

CPU = torch.device("cpu")


def config(P=8, tracks=("Drums", "Piano", "Guitar"), num_pixels=1):
    return {"model_name": "t", "data": {"pitch_range": {"lowest": 24, "highest": 24 + P}, "instruments": list(tracks), "beat_resolution": 4},
            "training": {"num_pixels": num_pixels, "random_seed": 23}}


def params(mode, enc="Pass", enc_hidden=None, gen="NADE", Hn=16, units=(32, 32), feedback=None, keep_prob=0.9):
    return {"mode": mode, "tune_encoder": False, "keep_prob": keep_prob, "encoder": {"type": enc, "num_hidden": enc_hidden},
            "generator": {"type": gen, "num_hidden": Hn, "num_hidden_rnn": list(units), "feedback": feedback}}


def test_reference_call_site_kwargs_construct_every_class():
    from multinn_amd import PassEncoder, DBNEncoder, RnnNade, RnnRBM, RnnMultiNADE
    # multinn_joint.py:44-48 / multi_encoder_nn.py:41-47: encoder_class(num_dims=, num_hidden=, track_name=)
    e = PassEncoder(num_dims=40, num_hidden=None, track_name="all")
    assert e.num_dims == 40 and e.num_hidden == [40] and e.num_layers == 1 and e.track_name == "all" and e.name == "pass-encoder"
    assert PassEncoder(num_dims=8, num_hidden=[168, 84], track_name="Piano").num_hidden == [8]        # num_hidden is unused (pass_encoder.py:26-29)
    d = DBNEncoder(num_dims=12, num_hidden=[10, 6], track_name="Drums", device=CPU)
    assert d.num_hidden == [10, 6] and d.num_layers == 2 and d.track_name == "Drums" and d.name == "dbn-encoder"
    assert d.dbn.rbms[0].k == 2 and d.dbn.rbm_layers[1].num_dims == 10                                  # dbn_encoder.py:22 default k = 2
    assert inspect.signature(DBNEncoder.__init__).parameters["k"].default == 2
    assert list(inspect.signature(PassEncoder.__init__).parameters)[1:5] == ["num_dims", "num_hidden", "name", "track_name"]
    assert list(inspect.signature(DBNEncoder.__init__).parameters)[1:6] == ["num_dims", "num_hidden", "k", "name", "track_name"]
    # multinn_joint.py:65-74: generator_class(num_dims=, num_hidden=, num_hidden_rnn=, keep_prob=)
    for cls, nm in ((RnnNade, "rnn-nade"), (RnnRBM, "rnn-rbm")):
        g = cls(num_dims=40, num_hidden=256, num_hidden_rnn=[512, 256], keep_prob=0.9)
        assert g.num_dims == 40 and g.num_hidden == [256] and g.num_hidden_rnn == [512, 256] and g.keep_prob == 0.9
        assert g.track_name == "all" and g.name == nm
    # multinn_jamming.py:40-48: + track_name=
    g = RnnRBM(num_dims=8, num_hidden=16, num_hidden_rnn=[32, 32], keep_prob=0.9, track_name="Bass")
    assert g.track_name == "Bass" and g.k == 10 and g.internal_bias is True                              # rnn_rbm.py:23-25 defaults
    assert RnnNade(num_dims=8, num_hidden=16, num_hidden_rnn=[32, 32], keep_prob=0.9, track_name="Bass").internal_bias is False
    # multinn_composer.py:49-57: RnnMultiNADE(num_dims=, num_hidden=, num_hidden_rnn=, tracks=, keep_prob=)
    m = RnnMultiNADE(num_dims=6, num_hidden=16, num_hidden_rnn=[32, 32], tracks=["a", "b", "c"], keep_prob=0.9)
    assert m.num_tracks == 3 and m.tracks == ["a", "b", "c"] and m.name == "rnn-multinade"


def test_encoder_lists_and_pass_encoder_semantics():
    from multinn_amd import PassEncoder
    e = PassEncoder(num_dims=4, num_hidden=None)
    x = torch.zeros(2, 3, 4)
    e.build(x)
    assert e.is_built and e.encodings[-1] is x and e.dec_probs[-1] is x and len(e.encodings) == 1      # pass_encoder.py:52-53: per-layer lists
    assert e.encode()[1] is x and e.decode(x)[0] is x
    assert e.train(None, 0.1) == ([], [], {}, [], {})                                                   # pass_encoder.py:129-136
    with pytest.raises(ValueError):
        e.build(x, mode="bogus")                                                                        # model.py:146-149
    with pytest.raises(ValueError):
        e.build_metrics(x, x)                                                                           # pass_encoder.py:86-88


@pytest.mark.parametrize("mode,cls_name", [("joint", "MultINNJoint"), ("composer", "MultINNComposer"), ("jamming", "MultINNJamming"),
                                           ("feedback", "MultINNFeedback"), ("feedback-rnn", "MultINNFeedbackRnn")])
def test_modes_wire_the_reference_classes(mode, cls_name):
    from multinn_amd import modes, PassEncoder, RnnNade, RnnMultiNADE
    P, tracks = 8, ("Drums", "Piano", "Guitar")
    m = modes.MultINN(config(P, tracks), params(mode, feedback=[32, 16]), mode=mode, device=CPU)
    assert type(m._model).__name__ == cls_name and m.mode == mode and m.num_tracks == 3 and m.num_dims == P
    assert m.keep_prob == 0.9 and m.tune_encoder is False and m.encoder_type == "Pass" and m.generator_type == "NADE"
    if mode == "joint":
        assert len(m.encoders) == 1 and m.encoders[0].num_dims == P * 3 and len(m.generators) == 1
        assert isinstance(m.generators[0], RnnNade) and m.generators[0].num_dims == P * 3               # multinn_joint.py:41-74
    elif mode == "composer":
        assert len(m.encoders) == 3 and len(m.generators) == 1 and isinstance(m.generators[0], RnnMultiNADE)
        assert m.generators[0].num_dims == P and m.generators[0].tracks == list(tracks)                  # multinn_composer.py:49-57
    else:
        assert len(m.encoders) == 3 and len(m.generators) == 3
        assert [g.track_name for g in m.generators] == list(tracks) and all(g.num_dims == P for g in m.generators)
        assert m.feedback_module == (mode.startswith("feedback"))
    assert all(isinstance(e, PassEncoder) for e in m.encoders)
    assert [e.track_name for e in m.encoders] == (["all"] if mode == "joint" else list(tracks))


def test_mode_argument_errors_follow_the_reference():
    from multinn_amd import modes
    with pytest.raises(ValueError):
        modes.MultINN(config(), params("joint"), mode="bogus")                                          # multinn.py:50-52
    with pytest.raises(ValueError):
        modes.MultINNJoint(config(), params("joint", enc="CNN"))                                        # multinn_core.py:46-47
    with pytest.raises(ValueError):
        modes.MultINNJoint(config(), params("joint", gen="GAN"))                                        # multinn_core.py:55-56
    with pytest.raises(NotImplementedError):
        modes.MultINNComposer(config(), params("composer", gen="RBM"))                                  # multinn_composer.py:44-45
    p = params("joint")
    p["tune_encoder"] = True                       # only lifts a stop_gradient nobody differentiates through (generator.py:201): accepted, same updates
    assert modes.MultINNJoint(config(), p).tune_encoder is True
    m = modes.MultINNJoint(config(P=8, num_pixels=3), params("joint"))
    assert m.num_dims == 24                                                                             # multinn_core.py:57-59
    with pytest.raises(ValueError):
        m.build(torch.zeros(2, 4, 8, 3, dtype=torch.uint8))                                             # wrong feature width
    with pytest.raises(ValueError):
        m.build(torch.zeros(2, 4, 24, 3, dtype=torch.uint8), mode="bogus")


def test_flatten_and_dbn_encoder_defaults():
    from multinn_amd.modes import flatten_maybe_padded_sequences
    x = torch.arange(2 * 3 * 2).reshape(2, 3, 2)
    assert torch.equal(flatten_maybe_padded_sequences(x), x.reshape(6, 2))
    got = flatten_maybe_padded_sequences(x, torch.tensor([2, 3]))                                       # b-major, then t (sequences.py:30-31)
    assert torch.equal(got, torch.tensor([[0, 1], [2, 3], [6, 7], [8, 9], [10, 11]]))


def test_no_cpu_fallback_in_the_mode_path():
    from multinn_amd import modes
    from multinn_amd._lib import MnnError
    if torch.cuda.is_available():
        pytest.skip("checks the CPU-only failure mode")
    m = modes.MultINNJoint(config(P=4, tracks=("a", "b")), params("joint", Hn=8, units=(32,)))
    with pytest.raises((MnnError, RuntimeError)):
        m.build(torch.zeros(2, 3, 4, 2, dtype=torch.uint8), None, True, "train")
