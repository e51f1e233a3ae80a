"""mmmot_amd - MI355X (gfx950) native forward path of the mmMOT tracking network.

Public surface mirrors the reference's ``modules`` package + ``build_model``:

    from mmmot_amd import TrackingNet, build_model
"""
from .modules import (AppearanceNet, NewEndIndicator_v2, PointNet_v1, SkipPool, TrackingNet,  # noqa: F401
                      affinity_module, fusion_module_A, fusion_module_B, fusion_module_C)

from .train import TrackingLoss  # noqa: E402,F401  (reference cost.py:134-185; mmmot_amd/train.py)

__all__ = ['TrackingLoss', 'build_criterion', 'TrackingNet', 'AppearanceNet', 'PointNet_v1', 'SkipPool', 'NewEndIndicator_v2', 'affinity_module',
           'fusion_module_A', 'fusion_module_B', 'fusion_module_C', 'build_model', 'model_kwargs_from_config']


def model_kwargs_from_config(common):
    """``config['common']`` (dict from experiments/*/config.yaml) -> TrackingNet kwargs.

    Same key mapping as reference utils/build_util.py:62-83."""
    m = common['model']
    return dict(
        seq_len=common['sample_max_len'], score_arch=m['score_arch'], appear_arch=m['appear_arch'],
        appear_len=m['appear_len'], appear_skippool=m['appear_skippool'], appear_fpn=m['appear_fpn'],
        point_arch=m['point_arch'], point_len=m['point_len'], without_reflectivity=common['without_reflectivity'],
        softmax_mode=m['softmax_mode'], affinity_op=m['affinity_op'], end_arch=m['end_arch'],
        end_mode=m['end_mode'], test_mode=m['test_mode'], score_fusion_arch=m['score_fusion_arch'],
        neg_threshold=m['neg_threshold'], dropblock=common['dropblock'], use_dropout=common['use_dropout'])


def build_model(config):
    """Drop-in for reference ``utils.build_util.build_model`` (accepts an EasyDict or a plain dict)."""
    common = config if 'model' in config else config['common']
    return TrackingNet(**model_kwargs_from_config(common))


def build_criterion(loss_cfg):
    """Drop-in for reference ``utils.build_util.build_criterion`` (utils/build_util.py:147-155): ``config['common']['loss']``."""
    g = lambda k, d=None: (loss_cfg.get(k, d) if hasattr(loss_cfg, 'get') else getattr(loss_cfg, k, d))
    return TrackingLoss(smooth_ratio=g('smooth_ratio', 0), detloss_type=g('det_loss', 'bce'), det_ratio=g('det_ratio', 0.4),
                        trans_ratio=g('trans_ratio', 0.4), trans_last=g('trans_last', False),
                        linkloss_type=g('link_loss', 'l2_softmax'))
