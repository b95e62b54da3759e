"""Layer classes of the framework (import surface of the reference's nn/layers package)."""
from . import convolutional, layers, maxpool, upsample

BaseLayer, BaseLayerGPU = layers.BaseLayer, layers.BaseLayerGPU
Param, ParamPack = layers.Param, layers.ParamPack
Concat, Flatten, FullyConnected, Noop = layers.Concat, layers.Flatten, layers.FullyConnected, layers.Noop
Relu, LeakyRelu, Sigmoid = layers.Relu, layers.LeakyRelu, layers.Sigmoid
Convolutional2D = convolutional.Convolutional2D
Conv2DToBatchedFixedWidthed = convolutional.Conv2DToBatchedFixedWidthed
MaxPool2D = maxpool.MaxPool2D
Upsample2D = upsample.Upsample2D

__all__ = ['BaseLayer', 'BaseLayerGPU', 'Param', 'ParamPack', 'Concat', 'Flatten', 'FullyConnected', 'Noop',
           'Relu', 'LeakyRelu', 'Sigmoid', 'Convolutional2D', 'Conv2DToBatchedFixedWidthed', 'MaxPool2D',
           'Upsample2D']
