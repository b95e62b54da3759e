"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, graph
flattening / naming / shapes / weight I/O work without a GPU, and compute calls fail loudly."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'univer_hip.h')).read()
    return sorted(set(re.findall(r'\b(uocr_[a-z0-9_]+)\s*\(', text)))


def test_abi_library_exports_every_declared_symbol():
    from univer_ocr_amd.hip import lib as hiplib
    path = hiplib.lib_path()
    assert os.path.exists(path), 'run ./build.sh (or __graft_entry__.build()) first'
    lib = ctypes.CDLL(path)
    declared = header_symbols()
    assert len(declared) >= 48
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/univer_hip.h but not exported'
    assert sorted(hiplib.ABI_SYMBOLS) == declared, 'ctypes prototypes out of sync with the header'
    assert lib.uocr_abi_version() == 4


@pytest.mark.parametrize('net_name', ['Monochrome', 'Paragraph', 'Line', 'Char'])
def test_net_structure_matches_reference(net_name):
    """Layer / parameter names and counts of the flattened nets equal the reference's (golden)."""
    from univer_ocr_amd.my_model.model import NET_MAKERS
    g = load_golden(f'my_model_{net_name.lower()}')
    model = NET_MAKERS[net_name](tuple(int(v) for v in g['in_shape']))
    assert sorted(model.layers) == [str(s) for s in g['layer_names']]
    assert sorted(model.params()) == [str(s) for s in g['param_names']]
    expected = {'Monochrome': 305, 'Paragraph': 130, 'Line': 1518, 'Char': 801442}[net_name]
    assert model.count_parameters() == expected            # SURVEY.md section 5 probe
    assert model.pack is not None and model.pack.total >= expected


def test_weights_json_round_trip_and_pack_views(tmp_path):
    """model_weights.json format (train.py:132-141): {layer: {param: nested list}}, compact separators."""
    from univer_ocr_amd.my_model.model import make_monochrome
    np.random.seed(3)
    model = make_monochrome((1, 8, 8, 1))
    weights = model.get_weights()
    assert set(weights) == {'Monochrome/conv_1', 'Monochrome/conv_2'}
    assert np.array(weights['Monochrome/conv_1']['w']).shape == (3, 3, 1, 16)
    assert np.array(weights['Monochrome/conv_2']['b']).shape == (1,)
    path = tmp_path / 'model_weights.json'
    path.write_text(json.dumps(weights, separators=(',', ':')))
    other = make_monochrome((1, 8, 8, 1))
    other.set_weights(json.loads(path.read_text()))
    for name, p in model.params().items():
        assert np.array_equal(p.value.numpy(), other.params()[name].value.numpy())
    # parameters are views of the flat pack: writing the pack changes the param and vice versa
    pack = other.pack
    p, off, size = pack.entries[0]
    pack.value.t[off] = 123.0
    assert p.value.numpy().reshape(-1)[0] == 123.0
    p.value = np.zeros(p.value.shape)
    assert float(pack.value.t[off]) == 0.0
    # NaN / wrong shapes are skipped (layers.py:123-137)
    bad = {'Monochrome/conv_1': {'w': np.full((3, 3, 1, 16), np.nan).tolist(), 'b': [0.0] * 3}}
    before = other.params()['Monochrome/conv_1/w'].value.numpy().copy()
    other.set_weights(bad)
    assert np.array_equal(other.params()['Monochrome/conv_1/w'].value.numpy(), before)


def test_graph_flattening_and_shapes():
    from univer_ocr_amd.nn.layers import Concat, Convolutional2D, MaxPool2D
    from univer_ocr_amd.nn.models import Model, Sequential

    def sub(out_ch):
        return Sequential([Convolutional2D((2, 2), out_channels=out_ch), Convolutional2D((2, 2), out_channels=out_ch),
                           MaxPool2D((2, 2))])
    layers = {'row_1': sub(2), 'row_2': sub(3), 'concat_rows': Concat(), 'concat_inputs': Concat(),
              'row_inputs': sub(2), 'concat_all': Concat(), 'pool_1': MaxPool2D((2, 2)),
              'pool_2': MaxPool2D((2, 2)), 'conv_end': Convolutional2D((2, 2), out_channels=3)}
    relations = {'row_1': 0, 'row_2': 1, 'concat_rows': ['row_1', 'row_2'], 'concat_inputs': [0, 1],
                 'row_inputs': 'concat_inputs', 'concat_all': ['concat_rows', 'row_inputs'],
                 'pool_1': 'concat_all', 'pool_2': 'pool_1', 'conv_end': 'pool_2', 0: 'conv_end'}
    model = Model(layers, relations)
    assert model.inputs_count == 2 and model.outputs_count == 1
    assert 'row_1/0_Convolutional2D' in model.layers and 'row_inputs/2_MaxPool2D' in model.layers
    assert model.relations['concat_rows'] == ['row_1/2_MaxPool2D', 'row_2/2_MaxPool2D']
    assert model.relations['row_inputs/0_Convolutional2D'] == ['concat_inputs']
    model.initialize([(3, 18, 18, 3), (3, 18, 18, 3)])
    g = load_golden('graph_models')
    assert sorted(model.layers) == [str(s) for s in g['nested/layer_names']]
    assert model.get_output_shapes([(3, 18, 18, 3)] * 2) == [(3, 1, 1, 3)]
    assert model.layers['row_inputs/0_Convolutional2D'].in_channels == 6
    with pytest.raises(RecursionError):
        Model({'a': Concat(), 'b': Concat()}, {'a': ['b', 0], 'b': 'a', 0: 'a'}).initialize([(1, 2, 2, 1)])
    with pytest.raises(TypeError):
        Model([], {})
    with pytest.raises(TypeError):
        Sequential({})


def test_receptive_fields_of_monochrome():
    from univer_ocr_amd.my_model.model import make_monochrome
    model = make_monochrome((1, 8, 8, 1))
    rf = model.get_receptive_fields()
    assert rf['Monochrome/conv_1']['input 0']['cnt'] == (3, 3)
    assert rf['Monochrome/conv_2']['input 0'] == {'cnt': (5, 5), 'y': (-2, 2), 'x': (-2, 2),
                                                   'is_solid_y': True, 'is_solid_x': True}


def test_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from univer_ocr_amd.hip import HipError
    from univer_ocr_amd.my_model.model import make_monochrome
    from univer_ocr_amd.nn import CP
    model = make_monochrome((1, 8, 8, 1))
    with pytest.raises(HipError):
        model.predict(CP.copy(np.zeros((1, 8, 8, 1))))
    with pytest.raises(NotImplementedError):
        CP.use_cpu()
