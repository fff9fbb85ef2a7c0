"""oracle/iou_ref.py against vectors produced by the reference's own iou_loss definitions (tests/golden/iou)."""
import os

import numpy as np

from oracle import iou_ref

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'iou', 'iou.npz'))


def test_iou_losses_match_the_reference_functions():
    assert abs(iou_ref.iou_loss_opt_shape(Z['pred'], Z['target']) - float(Z['iou_loss_opt_shape'])) < 1e-6
    assert abs(iou_ref.iou_loss_train_reconstruction(Z['pred'], Z['target']) - float(Z['iou_loss_train_reconstruction'])) < 1e-6
    got = iou_ref.multiview_iou_loss(list(Z['views']), Z['targets_a'], Z['targets_b'])
    assert abs(got - float(Z['multiview_iou_loss'])) < 1e-6


def test_both_scripts_define_the_same_loss():
    assert abs(iou_ref.iou_loss_opt_shape(Z['pred'], Z['target']) - iou_ref.iou_loss_train_reconstruction(Z['pred'], Z['target'])) < 1e-9


def test_empty_prediction_and_empty_target():
    i, u = iou_ref.iou_sums(Z['pred'], Z['target'])
    assert i[1] == 0 and u[1] == Z['target'][1].sum()          # empty prediction: union = sum(target)
    assert i[3] == 0 and abs(u[3] - Z['pred'][3].sum(dtype=np.float64)) < 1e-3   # empty target: union = sum(prediction)
