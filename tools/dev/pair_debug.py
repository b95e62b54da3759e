#!/usr/bin/env python3
"""Debug aid: uocr_conv_pair_bwd against the oracle, with an error map."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import nn_oracle as O  # noqa: E402
from univer_ocr_amd.hip import lib as hiplib  # noqa: E402
from univer_ocr_amd.nn import CP, ops  # noqa: E402

CP.use_gpu(0)
CP.set_dtype('float32')
rt = CP.runtime()
for opt in sys.argv[1:]:
    k, v = opt.split('=')
    rt.set_option(k, int(v))


def run(shape, sigmoid=True, bias=True, pad1=0.0):
    n, h, w = shape
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal((n, h, w, 1))
    w1 = rng.standard_normal((3, 3, 1, 16)) * 0.4
    b1 = rng.standard_normal(16) * 0.3
    w2 = rng.standard_normal((3, 3, 16, 1)) * 0.2
    b2 = rng.standard_normal(1)
    alpha = 0.01
    z1 = O.conv2d_fwd(X, w1, b1, 1, 1, pad1, bias)
    a1 = O.leaky_relu_fwd(z1, alpha)
    z2 = O.conv2d_fwd(a1, w2, b2, 1, 1, 0.0, bias)
    ref_y = O.sigmoid_fwd(z2) if sigmoid else z2
    g = rng.standard_normal(ref_y.shape)
    gz2 = O.sigmoid_bwd(z2, g) if sigmoid else g
    ga1, ref_dw2, ref_db2 = O.conv2d_bwd(a1, w2, gz2, 1, 1, 0.0, bias)
    gz1 = O.leaky_relu_bwd(z1, ga1, alpha)
    ref_dx, ref_dw1, ref_db1 = O.conv2d_bwd(X, w1, gz1, 1, 1, pad1, bias)
    act2 = hiplib.ACT_SIGMOID if sigmoid else hiplib.ACT_NONE
    Xd, w1d, b1d, w2d, b2d, gd = (CP.copy(a) for a in (X, w1, b1, w2, b2, g))
    y = CP.copy(ref_y)
    dw1, db1, dw2, db2 = CP.zeros(w1.shape), CP.zeros(b1.shape), CP.zeros(w2.shape), CP.zeros(b2.shape)
    dx = ops.conv_pair_bwd(Xd, y, gd, w1d, b1d, w2d, dw1, db1, dw2, db2, pad1, bias, bias, alpha, act2,
                           need_dx=True, accumulate=False)
    print(f'shape {shape} sigmoid={sigmoid} bias={bias} pad1={pad1}')
    for name, got, ref in (('dx', dx, ref_dx), ('dw1', dw1, ref_dw1), ('db1', db1, ref_db1), ('dw2', dw2, ref_dw2),
                           ('db2', db2, ref_db2)):
        got = CP.asnumpy(got).astype(np.float64)
        err = np.abs(got - ref) / max(1e-30, np.max(np.abs(ref)))
        print(f'  {name:4s} rel_linf {err.max():.3e}')
        if name == 'dx' and err.max() > 1e-4:
            bad = err[..., 0] > 1e-4
            print('   bad images', np.unique(np.nonzero(bad)[0])[:10], 'rows', np.unique(np.nonzero(bad)[1])[:40],
                  'cols', np.unique(np.nonzero(bad)[2])[:80])
            b = np.nonzero(bad)
            for k in range(min(6, len(b[0]))):
                i, r, c = b[0][k], b[1][k], b[2][k]
                print(f'   [{i},{r},{c}] got {got[i, r, c, 0]:+.5f} ref {ref[i, r, c, 0]:+.5f}')
        if name in ('dw1', 'dw2') and err.max() > 1e-4:
            print('   got', np.round(got.reshape(9, 16)[:, :4], 4).tolist())
            print('   ref', np.round(ref.reshape(9, 16)[:, :4], 4).tolist())


for shape in ((1, 8, 64), (1, 4, 64), (2, 16, 512), (3, 37, 83), (1, 1, 1), (1, 70, 33)):
    run(shape)
run((2, 5, 200), False, False, 0.25)
