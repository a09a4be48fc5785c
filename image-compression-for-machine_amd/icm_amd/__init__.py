"""icm_amd: MI355X-native hot path of stm233/image-compression-for-machine's `cnn` (WACNN), `stf` and `stf6` codecs.

The package mirrors the reference's ``compressai`` interface for this path (same class names, constructor
signatures, state-dict keys and error behaviour):

    icm_amd.ops              ste_round, LowerBound, NonNegativeParametrizer                  (compressai/ops)
    icm_amd.layers           GDN, conv3x3, subpel_conv3x3, conv1x1, Win_noShift_Attention    (compressai/layers)
    icm_amd.entropy_models   EntropyBottleneck, GaussianConditional (+ update / compress)   (compressai/entropy_models)
    icm_amd.ans              RansEncoder / RansDecoder / pmf_to_quantized_cdf (host coder)   (compressai.ans, _CXX)
    icm_amd.models           CompressionModel, WACNN, SymmetricalTransFormer(3)              (compressai/models)
    icm_amd.zigzag           ZigzagSplits / ZigzagReverse                                    (models/stf6.py:654-762)
    icm_amd.zoo              models = {"cnn": WACNN, "stf": ..., "stf6": ...}                (compressai/zoo)
    icm_amd.losses           RateDistortionLoss                                              (train.py:44-76)
    icm_amd.trainer          data-parallel training step (RCCL gradient all-reduce)          (train.py:172-233)
    icm_amd.datasets         ImageFolder + crop / tensor transforms                          (compressai/datasets)
    icm_amd.train            training CLI (checkpoints, lr schedule, test epoch)             (train.py:290-530)
    icm_amd.eval_model       evaluation CLI (pad-to-64 inference, bpp / PSNR report)         (utils/eval_model)
    icm_amd.graphs           hipGraph capture of the eval forward

All numerics run in hand-written HIP kernels behind the C ABI of include/icm_hip.h (lib/libicm_hip.so); there is no
CPU or ATen compute fallback: importing works without the library, any op raises if it is missing.
"""
__version__ = "0.3.0"
