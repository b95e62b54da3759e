"""Loss functions (reference: nn/losses.py:4-73): __call__(prediction, ground_truth) -> (loss, grad).

`loss` is a Python float (one device->host sync, as the reference's float(loss)) unless
CP.lazy_losses is set, in which case it is a DeviceScalar that syncs when converted."""
from . import ops


class BaseLoss:
    def __call__(self, prediction, ground_truth):
        raise NotImplementedError()

    def value_only(self, prediction, ground_truth):
        """Loss without the gradient tensor (Model.test, models.py:256-268)."""
        return self(prediction, ground_truth)[0]


class SegmentationDice2D(BaseLoss):
    folds_sigmoid = True      # __call__(..., out_act='sigmoid'): gradient w.r.t. the input of a fused output Sigmoid

    def __call__(self, prediction, ground_truth, need_grad=True, out_act=None):
        return ops.seg_loss('dice', ops.as_device(prediction), ops.as_device(ground_truth), need_grad, out_act)

    def value_only(self, prediction, ground_truth):
        return self(prediction, ground_truth, need_grad=False)[0]


class SegmentationJaccard2D(BaseLoss):
    folds_sigmoid = True

    def __call__(self, prediction, ground_truth, need_grad=True, out_act=None):
        return ops.seg_loss('jaccard', ops.as_device(prediction), ops.as_device(ground_truth), need_grad, out_act)

    def value_only(self, prediction, ground_truth):
        return self(prediction, ground_truth, need_grad=False)[0]


class SigmoidCrossEntropy(BaseLoss):
    def __call__(self, prediction, ground_truth, need_grad=True):
        return ops.sigmoid_ce(ops.as_device(prediction), ops.as_device(ground_truth), need_grad)

    def value_only(self, prediction, ground_truth):
        return self(prediction, ground_truth, need_grad=False)[0]


class SoftmaxCrossEntropy(BaseLoss):
    def __call__(self, prediction, ground_truth, need_grad=True):
        return ops.softmax_ce(ops.as_device(prediction), ops.as_device(ground_truth), need_grad)

    def value_only(self, prediction, ground_truth):
        return self(prediction, ground_truth, need_grad=False)[0]
