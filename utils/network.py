"""Drop-in module path of the reference (`from utils.network import ...`): gfx950 implementation."""
from nvfpcc_amd.network import *  # noqa: F401,F403
from nvfpcc_amd.network import (SingleLayerLatentGen, QuantGaussianLikelihood, CompDecoder, QConvTranspose3d,  # noqa: F401
                                QConv3d, IConv3d, GaussianLikelihoodModel, GDN3d, IGDN3d, reset_seed, set_noise_seed)
