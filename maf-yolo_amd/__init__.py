"""MAF-YOLO hot path, MI355X-native: Model.forward() / non_max_suppression() over hand-written HIP kernels.

See DESIGN.md.  Importing this package does not need a GPU; running it does (no CPU fallback)."""
from . import lib, arch, pack, synth  # noqa: F401
from .lib import MafError  # noqa: F401
from .model import Model, Detect_yaml  # noqa: F401
from .nms import non_max_suppression, non_max_suppression_async, nms_raw  # noqa: F401
from .post import convert_to_coco_format, coco_rows  # noqa: F401
from .checkpoint import load_checkpoint, reference_state_dict  # noqa: F401
from .loss import ComputeLoss, task_aligned_assign  # noqa: F401
from .streams import concurrent_streams  # noqa: F401
from .eval_loop import EvalLoop  # noqa: F401
from .solver import build_optimizer, ModelEMA, GradScaler  # noqa: F401
from .exchange import GradExchange  # noqa: F401
from .layers import RepVGGBlock, UniRepLKNetBlock  # noqa: F401  (isinstance loops of evaler.py:101-109 stay harmless)


def build_model(cfg, num_classes=80, device="cuda", img_size=640):
    """Counterpart of yolov6/models/yolo.py:290-297."""
    return Model(cfg, channels=3, num_classes=num_classes).to(device)
