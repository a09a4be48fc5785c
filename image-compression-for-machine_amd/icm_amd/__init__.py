"""icm_amd: MI355X-native hot path of stm233/image-compression-for-machine's `cnn` (WACNN) codec.

The package mirrors the reference's ``compressai`` interface for this path (same class names, constructor
signatures, state-dict keys and error behaviour):

    icm_amd.ops              ste_round, LowerBound, NonNegativeParametrizer      (compressai/ops)
    icm_amd.layers           GDN, conv3x3, subpel_conv3x3, conv1x1, Win_noShift_Attention (compressai/layers)
    icm_amd.entropy_models   EntropyBottleneck, GaussianConditional               (compressai/entropy_models)
    icm_amd.models           CompressionModel, WACNN                              (compressai/models)
    icm_amd.zoo              models = {"cnn": WACNN}                               (compressai/zoo)
    icm_amd.losses           RateDistortionLoss                                   (train.py:44-76)
    icm_amd.trainer          data-parallel training step (RCCL)                   (train.py:172-233)

All numerics run in hand-written HIP kernels (libicm_hip.so); there is no CPU or ATen fallback.
"""
__version__ = "0.1.0"
