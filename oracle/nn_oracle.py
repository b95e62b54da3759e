"""CPU ORACLE for the univer-ocr nn hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the arithmetic of the reference's self-written
deep-learning framework (KerkDovan/univer-ocr, web_app/components/nn).  Every
function cites the reference file:line it restates (paths relative to
/root/reference/web_app/components/nn/).  The reference computes each op with
Python loops over output pixels; this restatement computes the same sums
vectorised (im2col / strided slices), so results agree to float64 rounding
(checked to <=1e-12 against the golden vectors in tests/golden/, which were
produced by the reference itself -- tests/golden/make_golden.py).

PARITY PINNED: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz (reference outputs) including the reference's own
known-answer vectors (test/test_gradients.py:171-188, 216-222).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The shipped package never does.
"""
import math

import numpy as np

EPS_LOSS = 1e-8   # losses.py:17,36
EPS_OPT = 1e-8    # optimizers.py:5


def _pair(v):
    return (v, v) if isinstance(v, (int, np.integer)) else tuple(int(t) for t in v)


# ----------------------------------------------------------------------------
# Convolutional2D  (layers/convolutional.py)
# ----------------------------------------------------------------------------
def conv2d_out_hw(h, w, ks, stride, padding):
    """convolutional.py:290-301."""
    (kh, kw), (sh, sw), (ph, pw) = _pair(ks), _pair(stride), _pair(padding)
    oh = math.floor((h + 2 * ph - (kh - 1) - 1) / sh + 1)
    ow = math.floor((w + 2 * pw - (kw - 1) - 1) / sw + 1)
    return oh, ow


def _pad_const(X, ph, pw, value):
    """convolutional.py:79-83: border filled with padding_value."""
    if ph == 0 and pw == 0:
        return X
    b, h, w, c = X.shape
    out = np.full((b, h + 2 * ph, w + 2 * pw, c), float(value), dtype=X.dtype)
    out[:, ph:ph + h, pw:pw + w, :] = X
    return out


def _patches(Xp, kh, kw, sh, sw, oh, ow):
    """View (B, OH, OW, kh, kw, C) of the input windows (convolutional.py:92-93)."""
    b, hp, wp, c = Xp.shape
    s = Xp.strides
    return np.lib.stride_tricks.as_strided(
        Xp, shape=(b, oh, ow, kh, kw, c),
        strides=(s[0], s[1] * sh, s[2] * sw, s[1], s[2], s[3]), writeable=False)


def conv2d_fwd(X, w, b, stride=1, padding=0, padding_value=0.0, bias=True):
    """convolutional.py:62-99.  y = [patch, bias_flag] . [[w],[b]]."""
    kh, kw, cin, cout = w.shape
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    oh, ow = conv2d_out_hw(X.shape[1], X.shape[2], (kh, kw), (sh, sw), (ph, pw))
    Xp = _pad_const(X, ph, pw, padding_value)
    cols = _patches(Xp, kh, kw, sh, sw, oh, ow).reshape(X.shape[0] * oh * ow, kh * kw * cin)
    y = cols @ w.reshape(kh * kw * cin, cout)
    if bias:
        y = y + b
    return y.reshape(X.shape[0], oh, ow, cout)


def conv2d_bwd(X, w, g, stride=1, padding=0, padding_value=0.0, bias=True):
    """convolutional.py:101-145.  Returns (dx, dw, db).

    dw includes the contribution of the padded border (padding_value != 0 matters,
    :124-128 works on the padded X), db = sum(g) only when the bias flag is set
    (bias_vec = bias * ones, :113,125), dx is cropped back to the unpadded shape (:140-141).
    """
    kh, kw, cin, cout = w.shape
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    bsz, h, wd, _ = X.shape
    oh, ow = g.shape[1], g.shape[2]
    Xp = _pad_const(X, ph, pw, padding_value)
    cols = _patches(Xp, kh, kw, sh, sw, oh, ow).reshape(bsz * oh * ow, kh * kw * cin)
    g2 = g.reshape(bsz * oh * ow, cout)
    dw = (cols.T @ g2).reshape(kh, kw, cin, cout)
    db = g2.sum(axis=0) if bias else np.zeros((cout,), dtype=g.dtype)
    dcols = (g2 @ w.reshape(kh * kw * cin, cout).T).reshape(bsz, oh, ow, kh, kw, cin)
    dxp = np.zeros(Xp.shape, dtype=g.dtype)
    for ky in range(kh):
        for kx in range(kw):
            dxp[:, ky:ky + sh * oh:sh, kx:kx + sw * ow:sw, :] += dcols[:, :, :, ky, kx, :]
    dx = dxp[:, ph:ph + h, pw:pw + wd, :]
    return np.ascontiguousarray(dx), dw, db


# ---- the same two functions in the reference's own COST MODEL: one Python iteration per output pixel ----
# (convolutional.py:90-96 and :121-134).  Used by bench.py's cpu_baseline leg to time what the reference's
# NumPy path costs per page beside the vectorised restatement above; results agree with conv2d_fwd / conv2d_bwd
# to float64 rounding (tests/test_oracle_golden.py).
def conv2d_fwd_loop(X, w, b, stride=1, padding=0, padding_value=0.0, bias=True):
    """convolutional.py:62-99, loop for loop: per (y, x) a reshape, a concatenate with the bias column and a
    dot with [[w],[b]]."""
    kh, kw, cin, cout = w.shape
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    bsz = X.shape[0]
    oh, ow = conv2d_out_hw(X.shape[1], X.shape[2], (kh, kw), (sh, sw), (ph, pw))
    wb = np.concatenate((w.reshape(kh * kw * cin, cout), b.reshape(1, cout)))
    Xp = _pad_const(X, ph, pw, padding_value)
    result = np.zeros((bsz, oh, ow, cout))
    bias_vec = float(bool(bias)) * np.ones((bsz, 1))
    for y in range(oh):
        for x in range(ow):
            in_y, in_x = y * sh, x * sw
            i = np.reshape(Xp[:, in_y:in_y + kh, in_x:in_x + kw, :], (bsz, kh * kw * cin))
            i = np.concatenate((i, bias_vec), axis=1)
            result[:, y, x, :] = np.dot(i, wb)
    return result


def conv2d_bwd_loop(X, w, g, stride=1, padding=0, padding_value=0.0, bias=True):
    """convolutional.py:101-145, loop for loop: per (y, x) two dots and a scatter-add.  Returns (dx, dw, db)."""
    kh, kw, cin, cout = w.shape
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    bsz, h, wd, _ = X.shape
    oh, ow = g.shape[1], g.shape[2]
    Xp = _pad_const(X, ph, pw, padding_value)
    bias_vec = float(bool(bias)) * np.ones((bsz, 1))
    dx_total = np.zeros(Xp.shape)
    dw_temp = np.zeros((kh * kw * cin + 1, cout))
    wt = np.transpose(w.reshape(kh * kw * cin, cout))
    for y in range(oh):
        for x in range(ow):
            in_y, in_x = y * sh, x * sw
            cur = g[:, y, x, :]
            xt = np.reshape(Xp[:, in_y:in_y + kh, in_x:in_x + kw, :], (bsz, kh * kw * cin))
            xt = np.transpose(np.concatenate((xt, bias_vec), axis=1))
            dw_temp += np.dot(xt, cur)
            dx = np.reshape(np.dot(cur, wt), (bsz, kh, kw, cin))
            dx_total[:, in_y:in_y + kh, in_x:in_x + kw, :] += dx
    db = dw_temp[-1, :]
    dw = dw_temp[:-1, :].reshape(kh, kw, cin, cout)
    dx = dx_total[:, ph:ph + h, pw:pw + wd, :]
    return np.ascontiguousarray(dx), dw, db


# ----------------------------------------------------------------------------
# MaxPool2D  (layers/maxpool.py, CPU path = the oracle)
# ----------------------------------------------------------------------------
def maxpool2d_out_hw(h, w, ks, stride, padding, ceil_mode=False):
    """maxpool.py:204-216."""
    (kh, kw), (sh, sw), (ph, pw) = _pair(ks), _pair(stride), _pair(padding)
    rnd = math.ceil if ceil_mode else math.floor
    return (rnd((h + 2 * ph - (kh - 1) - 1) / sh + 1),
            rnd((w + 2 * pw - (kw - 1) - 1) / sw + 1))


def maxpool2d_fwd(X, ks, stride=None, padding=0, ceil_mode=False):
    """maxpool.py:24-58.  Returns (y, mask); mask is window-major (B, kh*OH, kw*OW, C).

    Zero padding takes part in the max (:36-40); windows that run past the padded
    extent (ceil_mode) shrink (:47) and leave the rest of their mask cell at 0.
    """
    kh, kw = _pair(ks)
    sh, sw = (kh, kw) if stride is None else _pair(stride)
    ph, pw = _pair(padding)
    bsz, h, w, c = X.shape
    oh, ow = maxpool2d_out_hw(h, w, (kh, kw), (sh, sw), (ph, pw), ceil_mode)
    Xp = _pad_const(X, ph, pw, 0.0)
    hp, wp = Xp.shape[1], Xp.shape[2]
    vals = np.full((kh, kw, bsz, oh, ow, c), -np.inf)
    valid = np.zeros((kh, kw, oh, ow), dtype=bool)
    for ky in range(kh):
        ny = min(oh, max(0, (hp - 1 - ky) // sh + 1))
        for kx in range(kw):
            nx = min(ow, max(0, (wp - 1 - kx) // sw + 1))
            vals[ky, kx, :, :ny, :nx, :] = Xp[:, ky:ky + sh * ny:sh, kx:kx + sw * nx:sw, :][:, :ny, :nx]
            valid[ky, kx, :ny, :nx] = True
    y = vals.max(axis=(0, 1))
    hit = (vals == y[None, None]) & valid[:, :, None, :, :, None]
    mask = np.zeros((bsz, kh * oh, kw * ow, c))
    for ky in range(kh):
        for kx in range(kw):
            mask[:, ky::kh, kx::kw, :] = hit[ky, kx]
    return y, mask


def maxpool2d_bwd(g, mask, in_shape, ks, stride=None, padding=0):
    """maxpool.py:62-90.  The gradient of a window is split equally between its ties (:83)."""
    kh, kw = _pair(ks)
    sh, sw = (kh, kw) if stride is None else _pair(stride)
    ph, pw = _pair(padding)
    bsz, h, w, c = in_shape
    oh, ow = g.shape[1], g.shape[2]
    hp, wp = h + 2 * ph, w + 2 * pw
    cnt = np.zeros(g.shape)
    for ky in range(kh):
        for kx in range(kw):
            cnt += mask[:, ky::kh, kx::kw, :]
    share = g / cnt
    dxp = np.zeros((bsz, hp, wp, c))
    for ky in range(kh):
        ny = min(oh, max(0, (hp - 1 - ky) // sh + 1))
        for kx in range(kw):
            nx = min(ow, max(0, (wp - 1 - kx) // sw + 1))
            m = mask[:, ky::kh, kx::kw, :][:, :ny, :nx]
            dxp[:, ky:ky + sh * ny:sh, kx:kx + sw * nx:sw, :] += (share[:, :ny, :nx] * m)
    return np.ascontiguousarray(dxp[:, ph:ph + h, pw:pw + w, :])


# ----------------------------------------------------------------------------
# Upsample2D  (layers/upsample.py)
# ----------------------------------------------------------------------------
def upsample2d_fwd(X, scale):
    """upsample.py:21-25."""
    sy, sx = _pair(scale)
    return X.repeat(sy, axis=1).repeat(sx, axis=2)


def upsample2d_bwd(g, scale):
    """upsample.py:27-39: sum over each (sy, sx) block."""
    sy, sx = _pair(scale)
    b, h, w, c = g.shape
    return g.reshape(b, h // sy, sy, w // sx, sx, c).sum(axis=(2, 4))


# ----------------------------------------------------------------------------
# activations / simple layers  (layers/layers.py)
# ----------------------------------------------------------------------------
def relu_fwd(X):
    """layers.py:377-381 (mask is X >= 0)."""
    return X * (X >= 0)


def relu_bwd(X, g):
    """layers.py:383-385."""
    return g * (X >= 0)


def leaky_relu_fwd(X, alpha=0.01):
    """layers.py:390-398."""
    return X * ((X >= 0) + alpha * (X < 0))


def leaky_relu_bwd(X, g, alpha=0.01):
    """layers.py:400-402."""
    return g * ((X >= 0) + alpha * (X < 0))


def sigmoid_fwd(X):
    """layers.py:407-410."""
    return 1 / (1 + np.exp(-X))


def sigmoid_bwd(X, g):
    """layers.py:412-415: recomputed from the stashed input."""
    e = np.exp(-X)
    return g * e / (e + 1) ** 2


def dense_fwd(X, w):
    """layers.py:335-339: y = [X, 1] . w, bias = last row of w."""
    return X @ w[:-1] + w[-1]


def dense_bwd(X, w, g):
    """layers.py:341-347.  Returns (dx, dw) with dw of shape (n_in + 1, n_out)."""
    dx = g @ w[:-1].T
    dw = np.concatenate((X.T @ g, g.sum(axis=0, keepdims=True)), axis=0)
    return dx, dw


def fixed_width_fwd(X, width):
    """convolutional.py:335-349: sliding window of `width` columns over the zero-padded W axis."""
    b, h, w, c = X.shape
    hw = width // 2
    padded = np.zeros((b, h, w + width, c))
    padded[:, :, hw:hw + w, :] = X
    win = np.lib.stride_tricks.sliding_window_view(padded, width, axis=2)   # (b,h,w+1,c,width)
    y = win[:, :, :w].transpose(0, 2, 1, 4, 3).reshape(b * w, h, width, c)
    return np.ascontiguousarray(y)


def fixed_width_bwd(g, in_shape, width):
    """convolutional.py:351-362 (scatter-add, then crop the padding)."""
    b, h, w, c = in_shape
    hw = width // 2
    dxp = np.zeros((b, h, w + width, c))
    g5 = g.reshape(b, w, h, width, c)
    for j in range(width):
        dxp[:, :, j:j + w, :] += g5[:, :, :, j, :].transpose(0, 2, 1, 3)
    return np.ascontiguousarray(dxp[:, :, hw:hw + w, :])


def concat_fwd(inputs, axis=-1):
    """layers.py:246-252."""
    return np.concatenate(inputs, axis=axis)


def concat_bwd(g, shapes, axis=-1):
    """layers.py:254-268."""
    sizes = np.cumsum([s[axis] for s in shapes])[:-1]
    return [np.ascontiguousarray(p) for p in np.split(g, sizes, axis=axis)]


# ----------------------------------------------------------------------------
# losses  (losses.py) -- each returns (float loss, grad)
# ----------------------------------------------------------------------------
def dice_loss(pred, gt):
    """losses.py:9-25."""
    num = (pred * gt).sum(axis=(1, 2), keepdims=True) + EPS_LOSS
    den = pred.sum(axis=(1, 2), keepdims=True) + gt.sum(axis=(1, 2), keepdims=True) + 2 * EPS_LOSS
    loss = np.sum(1 - 2 * num / den)
    grad = -2 * (gt * den - num) / den ** 2
    return float(loss), grad


def jaccard_loss(pred, gt):
    """losses.py:28-42."""
    num = (pred * gt).sum(axis=(1, 2), keepdims=True) + EPS_LOSS
    den = (pred.sum(axis=(1, 2), keepdims=True) + gt.sum(axis=(1, 2), keepdims=True)
           - num + 2 * EPS_LOSS)
    loss = np.sum(1 - num / den)
    grad = -(gt * den - num * (1 - gt)) / den ** 2
    return float(loss), grad


def sigmoid_ce_loss(pred, gt):
    """losses.py:45-57."""
    n = gt.shape[0]
    p = 1 / (1 + np.exp(-pred))
    loss = -np.sum(gt * np.log(p) + (1 - gt) * np.log(1 - p)) / n
    grad = (gt * (p - 1) + (1 - gt) * p) / n
    return float(loss), grad


def softmax_ce_loss(pred, gt):
    """losses.py:60-73 (max-subtracted softmax, no epsilon in the log)."""
    n = gt.shape[0]
    e = np.exp(pred - pred.max(axis=1, keepdims=True))
    p = e / e.sum(axis=1, keepdims=True)
    loss = -np.sum(gt * np.log(p)) / n
    return float(loss), (p - gt) / n


LOSSES = {'dice': dice_loss, 'jaccard': jaccard_loss,
          'sigmoid_ce': sigmoid_ce_loss, 'softmax_ce': softmax_ce_loss}


# ----------------------------------------------------------------------------
# regularizers (regularizations.py) and optimizers (optimizers.py)
# ----------------------------------------------------------------------------
def l1_reg(w, strength):
    """regularizations.py:15-19."""
    return float(strength * np.sum(np.abs(w))), strength * np.sign(w)


def l2_reg(w, strength):
    """regularizations.py:22-26."""
    return float(strength * np.sum(w ** 2)), strength * 2 * w


class AdamState:
    """optimizers.py:47-64: no bias correction, eps outside the sqrt."""

    def __init__(self, lr=0.001, beta1=0.9, beta2=0.999):
        self.lr, self.beta1, self.beta2 = lr, beta1, beta2
        self.state = {}

    def update(self, key, value, grad):
        v, a = self.state.get(key, (0.0, 0.0))
        v = self.beta1 * v + (1 - self.beta1) * grad
        a = self.beta2 * a + (1 - self.beta2) * grad ** 2
        self.state[key] = (v, a)
        return value - self.lr / (np.sqrt(a) + EPS_OPT) * v


class MomentumState:
    """optimizers.py:67-81 (momentum=0 is the SGD of BASELINE config 3)."""

    def __init__(self, lr, momentum=0.0):
        self.lr, self.momentum = lr, momentum
        self.state = {}

    def update(self, key, value, grad):
        v = self.momentum * self.state.get(key, 0.0) - self.lr * grad
        self.state[key] = v
        return value + v


class RMSPropState:
    """optimizers.py:84-98."""

    def __init__(self, lr=0.01, rho=0.99):
        self.lr, self.rho = lr, rho
        self.state = {}

    def update(self, key, value, grad):
        a = self.rho * self.state.get(key, 0.0) + (1 - self.rho) * grad ** 2
        self.state[key] = a
        return value - self.lr / (np.sqrt(a) + EPS_OPT) * grad


# ----------------------------------------------------------------------------
# sequential nets: the train step of models.py:232-254 for a chain of layers
# ----------------------------------------------------------------------------
class Net:
    """A chain of layer specs with the semantics of Model.train (models.py:232-254):
    forward (grads start from zero, models.py:188), loss, backward, L2 regularisation
    added into the grads (layers.py:147-155, models.py:472-476), optimizer update of
    every param (models.py:273-277).

    spec entries (name, kind, cfg):
      conv   cfg: ks, cin, cout, stride, padding, padding_value, bias, l2
      dense  cfg: n_in, n_out, l2
      leaky  cfg: alpha        relu / sigmoid / flatten / noop: {}
      upsample cfg: scale      maxpool cfg: ks, stride, padding, ceil_mode
      fixed_width cfg: width
    """

    def __init__(self, spec, loss, loops=False):
        self.spec = spec
        self.loss = loss
        self.params = {}
        self.grads = {}
        # loops=True: convolutions run one Python iteration per output pixel, as the reference's NumPy path does
        self._conv_fwd = conv2d_fwd_loop if loops else conv2d_fwd
        self._conv_bwd = conv2d_bwd_loop if loops else conv2d_bwd

    def param_names(self):
        return sorted(self.params.keys())

    def forward(self, X, keep=False):
        stash = []
        for name, kind, cfg in self.spec:
            inp = X
            if kind == 'conv':
                X = self._conv_fwd(X, self.params[f'{name}/w'], self.params[f'{name}/b'], cfg['stride'],
                               cfg['padding'], cfg.get('padding_value', 0.0), cfg.get('bias', True))
            elif kind == 'dense':
                X = dense_fwd(X, self.params[f'{name}/w'])
            elif kind == 'leaky':
                X = leaky_relu_fwd(X, cfg['alpha'])
            elif kind == 'relu':
                X = relu_fwd(X)
            elif kind == 'sigmoid':
                X = sigmoid_fwd(X)
            elif kind == 'upsample':
                X = upsample2d_fwd(X, cfg['scale'])
            elif kind == 'maxpool':
                X, mask = maxpool2d_fwd(X, cfg['ks'], cfg.get('stride'), cfg.get('padding', 0),
                                        cfg.get('ceil_mode', False))
                inp = (inp.shape, mask)
            elif kind == 'fixed_width':
                X = fixed_width_fwd(X, cfg['width'])
                inp = inp.shape
            elif kind == 'flatten':
                X = X.reshape(X.shape[0], -1)
                inp = inp.shape
            elif kind == 'noop':
                pass
            else:
                raise ValueError(kind)
            if keep:
                stash.append(inp)
        return (X, stash) if keep else X

    def backward(self, g, stash):
        grads = {}
        for (name, kind, cfg), inp in zip(reversed(self.spec), reversed(stash)):
            if kind == 'conv':
                g, dw, db = self._conv_bwd(inp, self.params[f'{name}/w'], g, cfg['stride'], cfg['padding'],
                                       cfg.get('padding_value', 0.0), cfg.get('bias', True))
                grads[f'{name}/w'], grads[f'{name}/b'] = dw, db
            elif kind == 'dense':
                g, dw = dense_bwd(inp, self.params[f'{name}/w'], g)
                grads[f'{name}/w'] = dw
            elif kind == 'leaky':
                g = leaky_relu_bwd(inp, g, cfg['alpha'])
            elif kind == 'relu':
                g = relu_bwd(inp, g)
            elif kind == 'sigmoid':
                g = sigmoid_bwd(inp, g)
            elif kind == 'upsample':
                g = upsample2d_bwd(g, cfg['scale'])
            elif kind == 'maxpool':
                g = maxpool2d_bwd(g, inp[1], inp[0], cfg['ks'], cfg.get('stride'), cfg.get('padding', 0))
            elif kind == 'fixed_width':
                g = fixed_width_bwd(g, inp, cfg['width'])
            elif kind == 'flatten':
                g = g.reshape(inp)
        return g, grads

    def loss_and_grads(self, X, y):
        """compute_loss_and_gradients (models.py:232-248)."""
        pred, stash = self.forward(X, keep=True)
        loss, g = LOSSES[self.loss](pred, y)
        dx, grads = self.backward(g, stash)
        reg = 0.0
        for name, kind, cfg in self.spec:
            lam = cfg.get('l2') if kind in ('conv', 'dense') else None
            if lam:
                for pn in ([f'{name}/w', f'{name}/b'] if kind == 'conv' else [f'{name}/w']):
                    rl, rg = l2_reg(self.params[pn], lam)
                    grads[pn] = grads[pn] + rg
                    reg += rl
        self.grads = grads
        return {'output_losses': [loss], 'regularization_loss': reg}, pred, dx

    def train_step(self, X, y, optimizer):
        losses, pred, _ = self.loss_and_grads(X, y)
        for pn, g in self.grads.items():
            self.params[pn] = optimizer.update(pn, self.params[pn], g)
        return losses, pred

    def test(self, X, y):
        pred = self.forward(X)
        loss, _ = LOSSES[self.loss](pred, y)
        return {'output_losses': [loss]}, pred


# my_model nets (my_model/model.py:108-304) as chains; names = unravelled layer names
def _conv_block(prefix, chans, cin, last_sigmoid, ks, padding, stride=1):
    """model.py:42-59 make_conv_block with make_conv's L2(0.01) (:36-39)."""
    spec = []
    for i, cout in enumerate(chans, 1):
        spec.append((f'{prefix}/conv_{i}', 'conv',
                     dict(ks=ks, cin=cin, cout=cout, stride=stride, padding=padding, l2=0.01)))
        if i == len(chans) and last_sigmoid:
            spec.append((f'{prefix}/sigmoid', 'sigmoid', {}))
        else:
            spec.append((f'{prefix}/leaky_relu_{i}', 'leaky', dict(alpha=0.01)))
        cin = cout
    return spec, cin


def monochrome_spec():
    """model.py:108-135."""
    spec, _ = _conv_block('Monochrome', [16, 1], 1, True, (3, 3), 1)
    return spec, 'dice'


def _unet_spec(root, width, out_ch):
    """model.py:138-191 (Paragraph, width 1) and :194-247 (Line, width 4)."""
    spec, c = [], 1
    for i in (1, 2):
        s, c = _conv_block(f'{root}/down_{i}', [width], c, False, (5, 5), 2, 2)
        spec += s
    for i in (2, 1):
        spec.append((f'{root}/up_{i}/upsample', 'upsample', dict(scale=2)))
        s, c = _conv_block(f'{root}/up_{i}/conv_block', [width], c, False, (5, 5), 2)
        spec += s
    s, c = _conv_block(f'{root}/end', [out_ch], c, True, (5, 5), 2)
    return spec + s


def paragraph_spec():
    return _unet_spec('Paragraph', 1, 1), 'dice'


def line_spec():
    return _unet_spec('Line', 4, 2), 'dice'


def char_spec(n_chars=162):
    """model.py:250-304: three 5x3 stride-(2,1) convs, fixed-width windows, three dense layers."""
    spec, c = _conv_block('Char/conv_block', [64, 64, 64], 1, False, (5, 3), (0, 1), (2, 1))
    spec.append(('Char/fixed_width', 'fixed_width', dict(width=8)))
    spec.append(('Char/flatten', 'flatten', {}))
    n_in = 8 * 64
    outs = [1024, 128, n_chars]
    for i, n_out in enumerate(outs, 1):
        spec.append((f'Char/dense_block/dense_{i}', 'dense', dict(n_in=n_in, n_out=n_out)))
        if i < len(outs):
            spec.append((f'Char/dense_block/leaky_relu_{i}', 'leaky', dict(alpha=0.01)))
        n_in = n_out
    return spec, 'softmax_ce'


NET_SPECS = {'Monochrome': monochrome_spec, 'Paragraph': paragraph_spec,
             'Line': line_spec, 'Char': char_spec}


def param_shapes(spec):
    shapes = {}
    for name, kind, cfg in spec:
        if kind == 'conv':
            kh, kw = _pair(cfg['ks'])
            shapes[f'{name}/w'] = (kh, kw, cfg['cin'], cfg['cout'])
            shapes[f'{name}/b'] = (cfg['cout'],)
        elif kind == 'dense':
            shapes[f'{name}/w'] = (cfg['n_in'] + 1, cfg['n_out'])
    return shapes


def analytic_weights(shape, salt):
    """The RNG-free initial weights the golden generator used (tests/golden/make_golden.py)."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.float64)
    fan_in = max(1, n // shape[-1])
    vals = np.sin(idx * 0.618 + salt * 1.37) * np.cos(idx * 0.0173 + salt) / np.sqrt(fan_in)
    return vals.reshape(shape)


def analytic_net_weights(spec):
    """Initial weights as make_golden.set_analytic_weights assigns them: salt = index of the
    layer in the sorted list of all layer names, +0.5 per param in sorted param order."""
    names = sorted(name for name, _, _ in spec)
    shapes = param_shapes(spec)
    out = {}
    for salt, lname in enumerate(names):
        pns = sorted(p for p in shapes if p.rsplit('/', 1)[0] == lname)
        for j, pn in enumerate(pns):
            out[pn] = analytic_weights(shapes[pn], salt + 0.5 * j)
    return out


def make_net(name, weights=None, loops=False):
    spec, loss = NET_SPECS[name]()
    net = Net(spec, loss, loops=loops)
    net.params = dict(analytic_net_weights(spec) if weights is None else weights)
    return net


def kaiming_uniform_weights(spec, rng):
    """initializers.py:22-25 semantics (sqrt(2/in) * U[0,1), non-negative) drawn from `rng`;
    conv w and b come from one (kh*kw*cin + 1, cout) draw (convolutional.py:41-45)."""
    out = {}
    for name, kind, cfg in spec:
        if kind == 'conv':
            kh, kw = _pair(cfg['ks'])
            n_in = kh * kw * cfg['cin'] + 1
            wb = rng.random((n_in, cfg['cout'])) / np.sqrt(n_in / 2)
            out[f'{name}/w'] = wb[:-1].reshape(kh, kw, cfg['cin'], cfg['cout'])
            out[f'{name}/b'] = wb[-1].copy()
        elif kind == 'dense':
            n_in = cfg['n_in'] + 1
            out[f'{name}/w'] = rng.random((n_in, cfg['n_out'])) / np.sqrt(n_in / 2)
    return out
