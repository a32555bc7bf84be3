"""MI355X-native (gfx950) implementation of the U-Net hot path of rg2/DeepFluoroLabeling-IPCAI2020.

Import as ``dfl_amd`` (see dfl_amd.py at the repository root; this directory's name is not a Python identifier).
Module names mirror the reference's flat files (unet, dice, ncc, util, dataset, warm_restarts_lr).
"""
from . import _native
from .unet import UNet
from .dice import DiceLoss2D, DiceAndHeatMapLoss2D
from .ncc import ncc_2d
from .util import center_crop, get_device
from .warm_restarts_lr import WarmRestartLR
from .sgd import SGD
from . import parallel
from .parallel import DataParallel

__all__ = ['UNet', 'DiceLoss2D', 'DiceAndHeatMapLoss2D', 'ncc_2d', 'center_crop', 'get_device', 'WarmRestartLR', 'SGD',
           'DataParallel', 'parallel']
