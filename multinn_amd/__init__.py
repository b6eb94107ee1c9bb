"""multinn_amd -- MI355X-native LSTM-NADE / LSTM-RBM train and sampling step for MultINN piano-rolls.

Host API mirrors ilya16/MultINN's Encoder/Generator plugin classes; compute is hand-written HIP
behind the C ABI in include/multinn_hip.h (libmultinn_hip.so).  No CPU fallback.
"""
__all__ = ["RnnNade", "RnnMultiNADE", "RnnRBM", "PassEncoder", "DBNEncoder", "NADE", "RBM", "RNN", "DBN",
           "AdamOptimizer", "GradientDescentOptimizer", "MultINN", "MultINNJoint", "MultINNComposer", "MultINNJamming",
           "MultINNFeedback", "MultINNFeedbackRnn"]


def __getattr__(name):
    if name in ("RnnNade", "RnnMultiNADE", "RnnRBM", "RnnEstimatorStateTuple", "Generator", "RnnEstimator"):
        from . import generators
        return getattr(generators, name)
    if name in ("PassEncoder", "DBNEncoder", "Encoder"):
        from . import encoders
        return getattr(encoders, name)
    if name in ("NADE", "RBM", "RNN", "DBN", "Model", "ParamStore"):
        from . import common
        return getattr(common, name)
    if name in ("DNN", "FeedbackDnn", "FeedbackRnn", "FeedbackRnnSampler", "FeedbackSampler"):
        from . import feedback
        return getattr(feedback, name)
    if name in ("MultINN", "MultINNJoint", "MultINNComposer", "MultINNJamming", "MultINNFeedback", "MultINNFeedbackRnn", "MultINNCore",
                "MultIEncoderNN"):
        from . import modes
        return getattr(modes, name)
    if name in ("AdamOptimizer", "GradientDescentOptimizer", "compute_gradients"):
        from . import training
        return getattr(training, name)
    raise AttributeError(name)
