"""reevr_amd -- MI355X-native partitioned-convolution engine (drop-in for the convolution hot
path of tiagolr/reevr). The product is csrc/libreevr_amd.so (HIP kernels + C ABI,
include/reevr_amd/rvc.h); this package is the thin host-side mirror of the reference's
convolver classes used by the tests and bench.py."""
from .convolver import (KERNEL_NAMES, Convolver, ConvolverSet, FFTConvolver, RvcError,  # noqa: F401
                        StereoConvolver, TUNING_DEFAULTS, TwoStageFFTConvolver, set_tuning, stage_plan, tuning)

from .impulse import Impulse  # noqa: F401,E402

__version__ = "0.2.0"
