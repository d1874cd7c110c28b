"""bnn_hip — host side of the MI355X Bayes-by-backprop hot path.

`_lib`       ctypes binding of libbnn_hip.so (include/bnn_hip.h); raises if it is missing
`ops`        one function per C-ABI entry point, on torch device tensors
`functional` autograd bridges (forward = HIP kernels)
`engine`     batched-samples launcher + MC-sample sharding over torch.distributed
`runtime`    process-wide knobs: math mode, Philox seed / sample counter, sharding switch
`optim`      FusedAdam: torch.optim.Adam's update in one launch (F2)
`train`      GraphedTrainStep: zero_grad -> sample_elbo -> backward -> Adam as one hipGraph
`synth`      synthetic inputs with the reference's distributions (numpy only)
"""
from .runtime import get_math, manual_seed, set_host_eps, set_math, shard_samples  # noqa: F401
from ._lib import BnnHipError  # noqa: F401

__all__ = ["get_math", "set_math", "manual_seed", "set_host_eps", "shard_samples", "BnnHipError"]
