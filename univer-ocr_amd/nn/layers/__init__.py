from .convolutional import Conv2DToBatchedFixedWidthed, Convolutional2D  # noqa: F401
from .layers import (  # noqa: F401
    BaseLayer, BaseLayerGPU, Concat, Flatten, FullyConnected, LeakyRelu, Noop, Param, ParamPack, Relu,
    Sigmoid)
from .maxpool import MaxPool2D  # noqa: F401
from .upsample import Upsample2D  # noqa: F401
