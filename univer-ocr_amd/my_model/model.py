"""Builders of the four OCR nets (reference: my_model/model.py:26-304) on the HIP backend:
same function names, layer names ('Monochrome/conv_1', 'Paragraph/up_2/conv_block/conv_1', ...),
channel counts, kernel sizes and losses, so `model_weights.json` files are interchangeable.

  Monochrome  conv3x3(1->16) LeakyReLU conv3x3(16->1) Sigmoid, Dice               (model.py:108-135)
  Paragraph   2 x [conv5x5 s2] down, 2 x [upsample2 + conv5x5] up, conv5x5 Sigmoid, width 1, Dice (:138-191)
  Line        same topology, width 4, 2 output maps, Dice                          (:194-247)
  Char        3 x conv(5x3, stride (2,1), pad (0,1), 64 ch) LeakyReLU, fixed-width(8) windows,
              dense 512->1024->128->162, softmax cross-entropy                     (:250-304)

Every conv carries L2(0.01) (model.py:36-39).  The crop / rotate stages that sit between the nets
in the reference's model system are host code (interpreter/) and out of this backend's scope;
`make_model_system` builds the modes whose data path stays on the device.
"""
from enum import Enum

import numpy as np

from ..nn.gpu import CP
from ..nn.help_func import make_list_if_not
from ..nn.layers import (
    Concat, Conv2DToBatchedFixedWidthed, Convolutional2D, Flatten, FullyConnected, LeakyRelu, Sigmoid,
    Upsample2D)
from ..nn.losses import SegmentationDice2D, SoftmaxCrossEntropy
from ..nn.model_system import ModelComponent, ModelSystem, StringSelector
from ..nn.models import Model
from ..nn.optimizers import Adam
from ..nn.regularizations import L2

CHAR_INPUT_HEIGHT = 32
CHAR_FIXED_WIDTH = 8
N_CHARS = 162            # len(primitives.CHARS): tab, space, 66 Cyrillic, 10 digits, 52 Latin, 32 punctuation
OUTPUT_CHANNELS = {'monochrome': 1, 'paragraph': 1, 'line': 2}   # constants.py:19-29 LAYER_NAMES


def make_divisible_by(arr, y, x):
    """model.py:26-34 (host side): zero-pad H, W up to the next multiple -- always adds at least 1."""
    b, h, w, c = arr.shape
    add_y, add_x = y - h % y, x - w % x
    out = np.zeros((b, h + add_y, w + add_x, c))
    out[:, add_y // 2:add_y // 2 + h, add_x // 2:add_x // 2 + w, :] = arr
    return out


def make_conv(out_ch, kernel_size=(5, 5), padding=2, **kwargs):
    return Convolutional2D(kernel_size, out_channels=out_ch, padding=padding, regularizer=L2(0.01), **kwargs)


def make_conv_block(out_chs, last_sigmoid=False, **kwargs):
    out_chs = make_list_if_not(out_chs)
    layers, relations, prev = {}, {}, 0
    for i, out_ch in enumerate(out_chs, 1):
        conv_name = f'conv_{i}'
        layers[conv_name] = make_conv(out_ch, **kwargs)
        relations[conv_name] = prev
        if i == len(out_chs) and last_sigmoid:
            act_name, act = 'sigmoid', Sigmoid()
        else:
            act_name, act = f'leaky_relu_{i}', LeakyRelu(0.01)
        layers[act_name] = act
        relations[act_name] = conv_name
        prev = act_name
    relations[0] = prev
    return Model(layers, relations)


def make_up(out_chs, **kwargs):
    return Model(layers={'upsample': Upsample2D(2), 'concat': Concat(),
                         'conv_block': make_conv_block(out_chs, **kwargs)},
                 relations={'upsample': 1, 'concat': ['upsample', 0], 'conv_block': 'concat', 0: 'conv_block'})


def make_single_up(out_chs, **kwargs):
    return Model(layers={'upsample': Upsample2D(2), 'conv_block': make_conv_block(out_chs, **kwargs)},
                 relations={'upsample': 0, 'conv_block': 'upsample', 0: 'conv_block'})


def wrap(name, model, **kwargs):
    return Model(layers={name: model}, relations={name: 0, 0: name}, **kwargs)


def _defaults(optimizer):
    return {'optimizer': Adam(lr=1e-2) if optimizer is None else optimizer, 'trainable': True}


def make_monochrome(input_shape, optimizer=None):
    kwargs = _defaults(optimizer)
    block = make_conv_block([16, OUTPUT_CHANNELS['monochrome']], last_sigmoid=True, kernel_size=(3, 3),
                            padding=1, **kwargs)
    model = Model(layers={'Monochrome': block}, relations={'Monochrome': 0, 0: 'Monochrome'},
                  loss=SegmentationDice2D())
    model.initialize(input_shape)
    return model


def _make_unet(root, width, out_ch, input_shape, optimizer):
    kwargs = _defaults(optimizer)
    depth = 2
    layers = {}
    for i in range(1, depth + 1):
        layers[f'down_{i}'] = make_conv_block([width], kernel_size=(5, 5), padding=2, stride=2, **kwargs)
    for i in range(1, depth + 1):
        layers[f'up_{i}'] = make_single_up([width], kernel_size=(5, 5), padding=2, **kwargs)
    layers['end'] = make_conv_block([out_ch], last_sigmoid=True, kernel_size=(5, 5), padding=2, **kwargs)
    relations = {'down_1': 0}
    for i in range(1, depth):
        relations[f'down_{i + 1}'] = f'down_{i}'
    relations[f'up_{depth}'] = f'down_{depth}'
    for i in range(1, depth):
        relations[f'up_{i}'] = f'up_{i + 1}'
    relations['end'] = 'up_1'
    relations[0] = 'end'
    model = wrap(root, Model(layers=layers, relations=relations), loss=SegmentationDice2D())
    model.initialize(input_shape)
    return model


def make_paragraph(input_shape, optimizer=None):
    return _make_unet('Paragraph', 1, OUTPUT_CHANNELS['paragraph'], input_shape, optimizer)


def make_line(input_shape, optimizer=None):
    return _make_unet('Line', 4, OUTPUT_CHANNELS['line'], input_shape, optimizer)


def make_dense_block(out_counts, **kwargs):
    out_counts = make_list_if_not(out_counts)
    layers, relations, prev = {}, {}, 0
    for i, n_out in enumerate(out_counts, 1):
        name = f'dense_{i}'
        layers[name] = FullyConnected(n_output=n_out, **kwargs)
        relations[name] = prev
        prev = name
        if i < len(out_counts):
            act = f'leaky_relu_{i}'
            layers[act] = LeakyRelu(0.01)
            relations[act] = name
            prev = act
    relations[0] = prev
    return Model(layers, relations)


def make_char(input_shape, optimizer=None):
    batch_size, _, width, in_channels = input_shape
    kwargs = _defaults(optimizer)
    layers = {
        'conv_block': make_conv_block([64, 64, 64], kernel_size=(5, 3), padding=(0, 1), stride=(2, 1), **kwargs),
        'fixed_width': Conv2DToBatchedFixedWidthed(CHAR_FIXED_WIDTH),
        'flatten': Flatten(),
        'dense_block': make_dense_block([1024, 128, N_CHARS], **kwargs),
    }
    relations = {'conv_block': 0, 'fixed_width': 'conv_block', 'flatten': 'fixed_width',
                 'dense_block': 'flatten', 0: 'dense_block'}
    model = wrap('Char', Model(layers=layers, relations=relations), loss=SoftmaxCrossEntropy())
    model.initialize((batch_size, CHAR_INPUT_HEIGHT, width, in_channels))
    return model


NET_MAKERS = {'Monochrome': make_monochrome, 'Paragraph': make_paragraph, 'Line': make_line, 'Char': make_char}


class Modes(Enum):
    TRAIN_MONOCHROME = 0
    TRAIN_PARAGRAPH = 1
    TRAIN_LINE = 2
    TRAIN_CHAR = 3
    TRAIN_ALL = 4
    PREDICT = 5
    TRAIN_PAGE = 6        # this backend: all four nets on device-resident page / line batches


def make_context_maker(mode=Modes.PREDICT):
    """model.py:412-483: dataset layers -> context dict with every array moved to the device."""
    def to_gpu(arr):
        return CP.copy(arr)

    wanted = {
        Modes.TRAIN_MONOCHROME: {'monochrome_X': 'image', 'monochrome_y': 'monochrome'},
        Modes.TRAIN_PARAGRAPH: {'paragraph_X': 'monochrome', 'paragraph_y': 'paragraph'},
        Modes.TRAIN_PAGE: {'monochrome_X': 'image', 'monochrome_y': 'monochrome',
                           'paragraph_X': 'monochrome', 'paragraph_y': 'paragraph',
                           'line_X': 'monochrome', 'line_y': 'line',
                           'char_X': 'char_lines', 'char_y': 'char_labels'},
        Modes.PREDICT: {'monochrome_X': 'image'},
    }
    if mode not in wanted:
        raise NotImplementedError(
            f'{mode.name} needs the host crop/rotate stages of the reference (interpreter/), '
            f'which are outside the MI355X backend (SURVEY.md section 2, component 18)')
    mapping = wanted[mode]

    def make_context(dataset_get_func, args=(), kwargs={}):
        tags = sorted(set(mapping.values()))
        layers = dataset_get_func(*args, layer_tags=tags, **kwargs)
        return {label: to_gpu(layers[tag]) for label, tag in mapping.items()}
    return make_context


def make_model_system(input_shape, optimizer=None, progress_tracker=None, weights=None, mode=Modes.PREDICT,
                      char_input_shape=None):
    """model.py:486-717 for the device-resident modes.  Returns (model_system, models, names)."""
    plan = {
        Modes.TRAIN_MONOCHROME: ['Monochrome'],
        Modes.TRAIN_PARAGRAPH: ['Paragraph'],
        Modes.TRAIN_PAGE: ['Monochrome', 'Paragraph', 'Line', 'Char'],
    }
    if mode not in plan:
        raise NotImplementedError(
            f'{mode.name} chains the nets through host crop/rotate stages (interpreter/), which are '
            f'outside the MI355X backend; use TRAIN_MONOCHROME / TRAIN_PARAGRAPH / TRAIN_PAGE')
    components, models = [], {}
    for name in plan[mode]:
        shape = input_shape
        if name == 'Char':
            shape = char_input_shape or (input_shape[0], CHAR_INPUT_HEIGHT, 64, 1)
        model = NET_MAKERS[name](shape, optimizer)
        if progress_tracker is not None:
            model.init_progress_tracker(progress_tracker, name)
        if weights is not None:
            model.set_weights(weights)
        key = name.lower()
        components.append(ModelComponent(name, model, StringSelector(f'{key}_X', f'{key}_y', f'{key}_pred'),
                                         delist_result=True))
        models[name] = model
    return ModelSystem(components), models, list(plan[mode])
