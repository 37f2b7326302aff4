"""MI355X-native YOLOv4 hot path behind the mmdet-yolov4 registry surface.

Importing the package registers ``DarknetCSP``, ``YOLOV4Neck``/``YOLOV5Neck``,
``YOLOCSPHead``, ``SingleStageDetector``, ``YOLOV4AnchorGenerator``,
``YOLOV4BBoxCoder`` and the ``Mish`` activation under the reference's names.  Numbers
come from ``lib/libyv4_hip.so`` (C-ABI: ``include/yv4.h``); there is no CPU fallback.
"""
from . import _lib
from .registry import (ACTIVATION_LAYERS, ANCHOR_GENERATORS, BACKBONES, BBOX_CODERS, DETECTORS, HEADS, LOSSES,
                       MODELS, NECKS, Config, ConfigDict, Registry, build_anchor_generator, build_backbone,
                       build_bbox_coder, build_detector, build_from_cfg, build_head, build_neck)
from .bricks import Mish, build_activation_layer, build_norm_layer, wrap_fp16_model
from .ops import (MishFunction, batched_nms, get_nms_iou_form, mish_backward, mish_forward, multiclass_nms, nms,
                  set_nms_iou_form, set_deterministic, deterministic)
from .anchor_generator import YOLOAnchorGenerator, YOLOV4AnchorGenerator
from .bbox_coder import YOLOV4BBoxCoder
from .darknetcsp import (Bottleneck, BottleneckCSP, BottleneckCSP2, Conv, CSPStage, DarknetCSP, Focus, SPPV4,
                         SPPV4Stage, SPPV5, SPPV5Stage, BottleneckStage)
from .yolo_neck_csp import YOLOV4Neck, YOLOV5Neck
from .yolocsp_head import YOLOCSPHead
from .single_stage import SingleStageDetector, bbox2result
from .yolov3 import Darknet, DetectionBlock, ResBlock, YOLOBBoxCoder, YOLOV3, YOLOV3Head, YOLOV3Neck
from .plan import Plan
from .preprocess import FusedTestPipeline
from .augment import FusedTrainPipeline
from .apis import inference_detector, single_gpu_test, multi_gpu_test
from .eval_utils import (coco_test_annotation, evaluate_fast_bbox, EVAL_BREAKDOWN, EVAL_IOU_CALCULATOR, EVAL_MATCHER, FlexibleStatisticsEval, IOU2DCoCo,
                         MatcherCoCo, ScaleBreakdown, average_precision, eval_map_flexible, iou_coco, match_coco)

__all__ = [n for n in dir() if not n.startswith('_')]
