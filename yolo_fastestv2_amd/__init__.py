"""yolo_fastestv2_amd - MI355X-native (gfx950) forward detection path of
dog-qiuqiu/Yolo-FastestV2: ShuffleNetV2 backbone, LightFPN, decoupled heads,
anchor decode and class-aware NMS as hand-written HIP kernels behind a C ABI
(include/yfv2.h, libyfv2.so), with the reference's own Python surface on top:

    Detector(classes, anchor_num, load_param, export_onnx=False)   model/detector.py:8
    handel_preds(preds, cfg, device)                               utils/utils.py:303
    non_max_suppression(prediction, conf_thres, iou_thres, classes) utils/utils.py:232
    get_batch_statistics(outputs, targets, iou_threshold, device)   utils/utils.py:194  (evaluation's matching loop)
    compute_loss(preds, targets, cfg, device)                       utils/loss.py:130   (training loss + its gradient w.r.t. the logits)

There is no CPU / PyTorch fallback: importing works anywhere, running needs the
built libyfv2.so and an MI355X.
"""
from ._lib import LIB_PATH, Yfv2Error  # noqa: F401
from .engine import Engine, get_engine, unpack_detections  # noqa: F401
from .model.detector import Detector  # noqa: F401
from .utils.utils import (ap_per_class, compute_ap, evaluation, get_batch_statistics, handel_preds, load_datafile,  # noqa: F401
                          nms_with_indices, non_max_suppression)
from .utils.loss import compute_loss  # noqa: F401
from .utils.optim import SGD  # noqa: F401
from .weights import export_weights, random_state_dict  # noqa: F401
from .pipeline import DetectPipeline  # noqa: F401
from .sharded import average_gradients_, detect_sharded, gather_decoded, gather_detections, shard_range  # noqa: F401


def install(reference_detector_module=None, reference_utils_module=None, reference_loss_module=None):
    """Swap the three hot-path symbols of an already-imported reference checkout
    (``import model.detector, utils.utils``) for the MI355X implementations, leaving
    everything else (config parsing, datasets, metrics, drawing) untouched."""
    if reference_detector_module is not None:
        reference_detector_module.Detector = Detector
    if reference_utils_module is not None:
        reference_utils_module.handel_preds = handel_preds
        reference_utils_module.non_max_suppression = non_max_suppression
        reference_utils_module.get_batch_statistics = get_batch_statistics   # evaluation()'s matching loop
        reference_utils_module.evaluation = evaluation                       # the loop itself, device-resident between batches
    if reference_loss_module is not None:
        reference_loss_module.compute_loss = compute_loss                    # utils/loss.py:130 (loss end only)
