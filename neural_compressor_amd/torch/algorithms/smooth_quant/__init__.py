from .save_load import load, save
from .smooth_quant import SmoothQuantQuantizer
from .utility import Calibration, SQLinearWrapper, TorchSmoothQuant, W8A8Linear, cal_scale

__all__ = ["SmoothQuantQuantizer", "TorchSmoothQuant", "Calibration", "SQLinearWrapper", "W8A8Linear", "cal_scale", "save", "load"]
